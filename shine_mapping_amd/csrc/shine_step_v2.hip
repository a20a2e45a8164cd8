// shine_step_v2.hip — the fused SHINE training step on 16-point tiles: four waves per SIMD.
//
//   query    FeatureOctree.query_feature   model/feature_octree.py:199-244
//   decode   Decoder.sdf                   model/decoder.py:49-63
//   loss     sdf_bce_loss                  utils/loss.py:17-24
//   backward cur_loss.backward()           shine_batch.py:208-209 (closed form, SURVEY.md §8a math contract)
//
// Same algorithm and the same outputs as shine_step_v1.hip; what changes is the shape of the work a wave owns.  v1's
// 32-point tile needs ~237 VGPRs and 18 KB of LDS per wave, i.e. TWO waves per SIMD — and the step is bound by latency
// chains (hash slot -> corner ids -> rows; atomics; LDS hand-offs), not by any pipe: with two waves there is nothing to
// run while both wait.  Here a wave owns 16 points:
//   * lane = (pt = lane & 15, g = lane >> 4); activations [32 channels x 16 points] live as two v_mfma_f32_16x16x4_f32
//     accumulators (exact fp32): D[16 x 16] += A[16 x 4] . B[4 x 16], lane l: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
//     acc reg r = D[4 (l>>4) + r][l&15]  ->  lane (pt, g) holds channels 16 m + 4 g + r of its point (m = accumulator);
//   * 8 registers per activation instead of 16, 9 KB of LDS per wave instead of 18: <= 128 VGPRs, 16 waves per CU;
//   * the chaining trick carries over: k-step (m', r) of the next product contracts over channels {16 m' + 4 g + r}, so
//     the ReLU'd accumulator register (m', r) IS its B operand; A operands are pre-permuted once per workgroup (s_opA);
//   * the 8-row products (dF = W1^T d1, dW1) pad 8 -> 16 instead of 8 -> 32: 68 half-size MFMAs per 16 points
//     (= 34 full-size per 16, against 42 in v1);
//   * query: the four lanes of a point split its 8 corners (lane g gathers the full 32-B rows of corners 2g, 2g+1: still
//     1 KiB per gather instruction), accumulate partial sums over all levels, and one reduce-scatter over g (6 lane
//     exchanges per tile) leaves features (2g, 2g+1) in lane g — the B operand of layer 1;
//   * scatter, trash rows, partial-vector flush: as in v1 (lane = (corner, feature), run-length over the node-ordered
//     stream, one 64-lane atomic per node run).
// Workgroup = WAVES x 64 threads: 16 waves (one workgroup per CU) for large batches, 4 waves for small ones (N = 4096
// fills 64 CUs instead of 16).
#include "shine_step_common.hpp"

namespace shine {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int V2_TP = 16;                      // points per tile
constexpr int V2_WP = 16;                      // pitch of the [corner][point] staging rows
constexpr int V2_TT = 20;                      // transpose tile pitch (floats, 16-B aligned rows)
constexpr int V2_DFP = 20;                     // pitch of the [feature][point] rows
constexpr int V2_IDS = LCAP * 8 * V2_WP;       // ids [LCAP][8][16] int32
constexpr int V2_W = LCAP * 8 * V2_WP;         // w   [LCAP][8][16]
constexpr int V2_R2 = 2 * 32 * V2_TT;          // two transpose tiles [32][20]; second life: df / J / cq
constexpr int V2_DF = 0, V2_CQ = 16 * V2_DFP;
constexpr int V2_WAVE_FLOATS = V2_IDS + V2_W + V2_R2;  // 2304 floats = 9216 B per wave
constexpr int V2_OPA1 = 0, V2_OPA2 = 4 * 64, V2_OPA2T = 20 * 64, V2_OPA1T = 36 * 64, V2_OPTOTAL = 44 * 64;
constexpr int V2_BIG = 12;                     // waves per workgroup of the full-chip launch (V2_BIG / 4 per SIMD)
// 68 16x16x4 MFMAs per 16-point tile: 4 + 16, 16 + 8, 16 + 8

static_assert(V2_DFP == V2_TT, "f_wr addresses both the transpose rows and the df rows");
static_assert(V2_CQ + LCAP * 8 * V2_WP <= V2_R2, "df/J/cq must fit the transpose region");
static_assert(PART_STRIDE <= V2_WAVE_FLOATS, "each wave's partial vector aliases its staging region at the end");

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 zero4() {
  f32x4 z;
  z[0] = z[1] = z[2] = z[3] = 0.f;
  return z;
}

template <int L, int WAVES, bool PROF>
__global__ __launch_bounds__(WAVES * 64, WAVES == V2_BIG ? V2_BIG / 4 : 2) void k_step_v2(V1Args a) {
  constexpr int NT = WAVES * 64;
  __shared__ float s_opA[V2_OPTOTAL];
  __shared__ float s_bias[100];
  __shared__ double s_loss[4];
  __shared__ float s_wave[WAVES][V2_WAVE_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pt = lane & 15, g = lane >> 4;
  const bool poly = a.poly != 0;
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tk = PROF ? clk() : 0;
#define SHINE_STAMP(k)            \
  if (PROF) {                     \
    long long now__ = clk();      \
    pc[k] += now__ - tk;          \
    tk = now__;                   \
  }

  // ---- per-workgroup setup: A operands in 16x16x4 lane order (lane l: row i = l & 15, k = l >> 4), biases
  for (int idx = tid; idx < V2_OPTOTAL; idx += NT) {
    const int t = idx >> 6, l = idx & 63, i = l & 15, kg = l >> 4;
    float v;
    if (t < 4) {  // W1: M-block mb = t >> 1, k-step tt = t & 1 contracts over features 2 kg + tt
      v = a.mlp[0][(16 * (t >> 1) + i) * F + 2 * kg + (t & 1)];
    } else if (t < 20) {  // W2: mb, k-step (m', r) contracts over channels 16 m' + 4 kg + r
      const int u = t - 4, mb = u >> 3, ks = u & 7;
      v = a.mlp[2][(16 * mb + i) * H + 16 * (ks >> 2) + 4 * kg + (ks & 3)];
    } else if (t < 36) {  // W2^T
      const int u = t - 20, mb = u >> 3, ks = u & 7;
      v = a.mlp[2][(16 * (ks >> 2) + 4 * kg + (ks & 3)) * H + 16 * mb + i];
    } else {  // W1^T, output rows permuted: row 4 g' + r' = feature 2 g' + r' for r' < 2, zero otherwise
      const int ks = t - 36, gp = i >> 2, rp = i & 3;
      const float w = a.mlp[0][(16 * (ks >> 2) + 4 * kg + (ks & 3)) * F + 2 * gp + (rp & 1)];
      v = rp < 2 ? w : 0.f;
    }
    s_opA[idx] = v;
  }
  if (tid < 32) {
    s_bias[tid] = a.mlp[1][tid];
    s_bias[32 + tid] = a.mlp[3][tid];
    s_bias[64 + tid] = a.mlp[4][tid];
  }
  if (tid == 0) {
    s_bias[96] = a.mlp[5][0];
    s_loss[0] = s_loss[1] = s_loss[2] = s_loss[3] = 0.0;
  }
  __syncthreads();

  float* U = s_wave[wv];
  int* U_ids = reinterpret_cast<int*>(U);  // [LCAP][8][16]
  float* U_w = U + V2_IDS;                 // [LCAP][8][16]
  float* R2 = U + V2_IDS + V2_W;

  // Per-lane LDS base addresses: every staging access below is one of these + a compile-time offset (DS instructions
  // carry a 16-bit immediate).  They are re-derived from an opaque lane value at the top of every tile, which keeps
  // LLVM's loop-invariant code motion from parking ~30 pre-added address variants in VGPRs across the whole loop.
  int lane_o = lane;
  const float b3 = s_bias[96];
  const float sigma = a.sigma;
  // this lane's channels are 16 m + 4 g + r: its biases / w3 entries are two 16-B LDS reads each, re-read per tile
  // (24 registers held across the whole kernel cost a wave of occupancy)
  const float4* sb4 = reinterpret_cast<const float4*>(s_bias);

  f32x4 accW2[2][2], accW1[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    accW1[m] = zero4();
#pragma unroll
    for (int n = 0; n < 2; ++n) accW2[m][n] = zero4();
  }
  float dw3c[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) dw3c[r] = 0.f;
  float db2acc[2] = {0.f, 0.f};
  float db3 = 0.f;
  float loss_acc = 0.f;  // per-lane sum over this wave's <= a few dozen tiles; widened to double at the flush
  int cnt_acc = 0;
  int run_id[LCAP], last_slot[LCAP], run_hit[LCAP];
  float run_acc[LCAP];
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

  // software prefetch of the {perm -> coord, label, slots} chain, index two tiles ahead (as in v1)
  long long np = 0;
  float nx0 = 0.f, nx1 = 0.f, nx2 = 0.f, nlabel = 0.f;
  int nslot[LCAP];
#pragma unroll
  for (int s = 0; s < LCAP; ++s) nslot[s] = -1;
  bool nvalid = begin + pt < end;
  int np2 = 0;
  if (a.perm && begin + V2_TP + pt < end) np2 = a.perm[begin + V2_TP + pt];
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
  }
  SHINE_STAMP(0)  // setup

  for (long long base = begin; base < end; base += V2_TP) {
    asm volatile("" : "+v"(lane_o));  // opaque per tile (see above)
    const int o_pt = lane_o & 15, o_g = lane_o >> 4;
    int* const st_ids = U_ids + (2 * o_g) * V2_WP + o_pt;   // staging writes: + (s * 8 [+ 1]) * V2_WP
    float* const st_w = U_w + (2 * o_g) * V2_WP + o_pt;
    const int* const sc_ids = U_ids + (lane_o >> 3) * V2_WP;  // scatter reads: + s * 8 * V2_WP + point
    const float* const sc_w = U_w + (lane_o >> 3) * V2_WP;
    const float* const sc_df = R2 + V2_DF + (lane_o & 7) * V2_DFP;
    float* const t_wr = R2 + (4 * o_g) * V2_TT + o_pt;       // transpose writes: + (16 m + r) * V2_TT [+ 32 * V2_TT]
    const float* const t_rd = R2 + o_pt * V2_TT + 4 * o_g;   // operand reads (i16 = lane & 15, kk = lane >> 4)
    float* const f_wr = R2 + (2 * o_g) * V2_TT + o_pt;       // [feature 2g (+1)][pt] rows (V2_DFP == V2_TT)
    const float* const opa = s_opA + lane_o;
    const bool valid = nvalid;
    const long long p = np;
    const long long po = a.pool_mode ? base + pt : p;  // where this point's outputs go
    const float x0 = nx0, x1 = nx1, x2 = nx2, label = nlabel;
    // ================================================================ phase 1: query (all levels)
    unsigned int chgmask[LCAP], hitmask[LCAP];
    const unsigned int validmask = (unsigned int)__ballot(valid) & 0xFFFFu;
    int slot[LCAP];
    if (a.slots) {
#pragma unroll
      for (int s = 0; s < L; ++s) slot[s] = valid ? nslot[s] : -1;
    } else {  // no plan: hash and probe here (leaf Morton key once; a parent's code is the child's >> 3)
      const float rl = a.lv[L - 1].res;
      const unsigned long long kleaf = morton3(quantize(x0, rl), quantize(x1, rl), quantize(x2, rl));
#pragma unroll
      for (int s = 0; s < L; ++s) {
        LevelDev Lv = {};
        Lv.keys = a.lv[s].keys;
        Lv.shift = a.lv[s].shift;
        Lv.mask = a.lv[s].mask;
        const int sl = probe(Lv, kleaf >> (3 * (L - 1 - s)));
        slot[s] = valid ? sl : -1;
      }
    }
    int2 id2[LCAP];
#pragma unroll
    for (int s = 0; s < L; ++s) {  // the corner ids of every level in flight together (this lane: corners 2g, 2g+1)
      const unsigned int sl = slot[s] >= 0 ? (unsigned int)slot[s] : 0u;
      id2[s] = *reinterpret_cast<const int2*>(reinterpret_cast<const int*>(a.lv[s].vals) + 8u * sl + 2u * (unsigned int)o_g);
    }
    float pf[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) pf[q] = 0.f;
#pragma unroll
    for (int s = 0; s < L; ++s) {
      const V1Level& Lv = a.lv[s];
      const bool hit = slot[s] >= 0;
      // the two full rows of this lane's corners: 4 x 16 B; a miss reads row 0 with weight 0 (no branches)
      const unsigned int o0 = (hit ? (unsigned int)id2[s].x : 0u) * (unsigned int)F;
      const unsigned int o1 = (hit ? (unsigned int)id2[s].y : 0u) * (unsigned int)F;
      const float4 r00 = *reinterpret_cast<const float4*>(Lv.feat + o0);
      const float4 r01 = *reinterpret_cast<const float4*>(Lv.feat + o0 + 4u);
      const float4 r10 = *reinterpret_cast<const float4*>(Lv.feat + o1);
      const float4 r11 = *reinterpret_cast<const float4*>(Lv.feat + o1 + 4u);
      // node-run boundaries of the sorted stream (wave-uniform bit masks over the 16 points of the tile)
      int prev = __shfl_up(slot[s], 1, 64);
      if (pt == 0) prev = last_slot[s];
      chgmask[s] = (unsigned int)__ballot(valid && slot[s] != prev) & 0xFFFFu;
      hitmask[s] = (unsigned int)__ballot(hit) & 0xFFFFu;
      last_slot[s] = __builtin_amdgcn_readlane(slot[s], 15);
      // corner weights in the reference's association (model/feature_octree.py:186-193): (phi_x * phi_y) * phi_z
      const Axis X = axis_weight_rt(poly, x0, Lv.res), Y = axis_weight_rt(poly, x1, Lv.res),
                 Z = axis_weight_rt(poly, x2, Lv.res);
      const float ax = (o_g >> 1) ? X.t : __fsub_rn(1.0f, X.t);
      const float ay = (o_g & 1) ? Y.t : __fsub_rn(1.0f, Y.t);
      const float axy = __fmul_rn(ax, ay);
      float w0 = __fmul_rn(axy, __fsub_rn(1.0f, Z.t)), w1 = __fmul_rn(axy, Z.t);
      if (!valid || !hit) w0 = w1 = 0.f;  // padding lanes and misses contribute nothing to f or to the scatter
      const int mneg = hit ? 0 : -1;  // a miss stages -1 (trash row), never the speculative ids
      st_ids[(s * 8) * V2_WP] = id2[s].x | mneg;
      st_ids[(s * 8 + 1) * V2_WP] = id2[s].y | mneg;
      st_w[(s * 8) * V2_WP] = w0;
      st_w[(s * 8 + 1) * V2_WP] = w1;
      pf[0] = fmaf(w0, r00.x, pf[0]);
      pf[1] = fmaf(w0, r00.y, pf[1]);
      pf[2] = fmaf(w0, r00.z, pf[2]);
      pf[3] = fmaf(w0, r00.w, pf[3]);
      pf[4] = fmaf(w0, r01.x, pf[4]);
      pf[5] = fmaf(w0, r01.y, pf[5]);
      pf[6] = fmaf(w0, r01.z, pf[6]);
      pf[7] = fmaf(w0, r01.w, pf[7]);
      pf[0] = fmaf(w1, r10.x, pf[0]);
      pf[1] = fmaf(w1, r10.y, pf[1]);
      pf[2] = fmaf(w1, r10.z, pf[2]);
      pf[3] = fmaf(w1, r10.w, pf[3]);
      pf[4] = fmaf(w1, r11.x, pf[4]);
      pf[5] = fmaf(w1, r11.y, pf[5]);
      pf[6] = fmaf(w1, r11.z, pf[6]);
      pf[7] = fmaf(w1, r11.w, pf[7]);
      // at most two levels' rows in flight (16 registers each): four waves per SIMD hide the rest of the latency
      if (s & 1) __builtin_amdgcn_sched_barrier(0);
    }
    // prefetch of tile t+1's point data, issued after every gather of this tile (vmcnt counts in order)
    {
      const long long ni = base + V2_TP + pt;
      nvalid = ni < end;
      np = 0;
      nx0 = nx1 = nx2 = nlabel = 0.f;
#pragma unroll
      for (int s = 0; s < LCAP; ++s) nslot[s] = -1;
      if (nvalid) {
        np = a.perm ? (long long)np2 : ni;
        if (a.slots) {
          const long long si = a.pool_mode ? np : ni;
#pragma unroll
          for (int s = 0; s < L; ++s) nslot[s] = __builtin_nontemporal_load(a.slots + si * L + s);
        }
        nx0 = __builtin_nontemporal_load(a.coord + 3 * np);
        nx1 = __builtin_nontemporal_load(a.coord + 3 * np + 1);
        nx2 = __builtin_nontemporal_load(a.coord + 3 * np + 2);
        nlabel = __builtin_nontemporal_load(a.label + np);
      }
      if (a.perm && ni + V2_TP < end) np2 = __builtin_nontemporal_load(a.perm + ni + V2_TP);
    }
    // reduce-scatter of the partial sums over the point's four lanes: lane g ends with features (2g, 2g+1)
    float f2[2];
    {
      float h4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float send = g < 2 ? pf[4 + q] : pf[q];
        const float keep = g < 2 ? pf[q] : pf[4 + q];
        h4[q] = keep + __shfl_xor(send, 32, 64);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float send = (g & 1) ? h4[t] : h4[2 + t];
        const float keep = (g & 1) ? h4[2 + t] : h4[t];
        f2[t] = keep + __shfl_xor(send, 16, 64);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // phase boundary: no operand of the next phase is fetched early
    SHINE_STAMP(1)  // query

    // ================================================================ phase 2: decoder forward (MFMA chain)
    f32x4 c1[2], c2[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float4 v1b = sb4[4 * m + g], v2b = sb4[8 + 4 * m + g];
      c1[m][0] = v1b.x, c1[m][1] = v1b.y, c1[m][2] = v1b.z, c1[m][3] = v1b.w;
      c2[m][0] = v2b.x, c2[m][1] = v2b.y, c2[m][2] = v2b.z, c2[m][3] = v2b.w;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m) c1[m] = mfma16(opa[V2_OPA1 + (2 * m + t) * 64], f2[t], c1[m]);
    float h1[8], h2[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) h1[r] = fmaxf(c1[r >> 2][r & 3], 0.f);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int m = 0; m < 2; ++m) c2[m] = mfma16(opa[V2_OPA2 + (8 * m + ks) * 64], h1[ks], c2[m]);
    float yp = 0.f;
    float w3r[8];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float4 v = sb4[16 + 4 * m + g];
      w3r[4 * m] = v.x, w3r[4 * m + 1] = v.y, w3r[4 * m + 2] = v.z, w3r[4 * m + 3] = v.w;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      h2[r] = fmaxf(c2[r >> 2][r & 3], 0.f);
      yp = fmaf(w3r[r], h2[r], yp);
    }
    yp += __shfl_xor(yp, 16, 64);
    const float y = yp + __shfl_xor(yp, 32, 64) + b3;
    if (valid && g == 0 && a.pred) __builtin_nontemporal_store(y, a.pred + po);
    __builtin_amdgcn_sched_barrier(0);  // phase boundary: no operand of the next phase is fetched early
    SHINE_STAMP(2)  // decoder forward

    // ================================================================ phase 3: loss
    float delta = 0.f;
    if (valid) {
      const float zt = sigmoidf_acc(label / sigma);
      if (g == 0) {
        loss_acc += fmaxf(y, 0.f) - y * zt + log1pf(expf(-fabsf(y)));
        cnt_acc += 1;
      }
      delta = (sigmoidf_acc(y) - zt) * a.inv_n;
    }
    // ================================================================ phase 4: backward through the decoder
    float d2[8], d1[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      d2[r] = h2[r] > 0.f ? delta * w3r[r] : 0.f;
      dw3c[r] = fmaf(delta, h2[r], dw3c[r]);
    }
    if (g == 0) db3 += delta;
    f32x4 e1[2] = {zero4(), zero4()}, e0 = zero4();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int m = 0; m < 2; ++m) e1[m] = mfma16(opa[V2_OPA2T + (8 * m + ks) * 64], d2[ks], e1[m]);
#pragma unroll
    for (int r = 0; r < 8; ++r) d1[r] = h1[r] > 0.f ? e1[r >> 2][r & 3] : 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) e0 = mfma16(opa[V2_OPA1T + ks * 64], d1[ks], e0);
    const float df2[2] = {e0[0], e0[1]};  // d loss / d f for features 2g, 2g+1 of this lane's point
    __builtin_amdgcn_sched_barrier(0);  // phase boundary: no operand of the next phase is fetched early
    SHINE_STAMP(3)  // loss + decoder backward

    // ================================================================ phase 5: decoder weight grads (transposed MFMA)
    if (a.decoder_grad_on) {
      const int i16 = lane & 15;  // operand role: row / column i16 = lane & 15, points 4 kk .. 4 kk + 3 (kk = lane >> 4)
#pragma unroll
      for (int r = 0; r < 8; ++r) {  // channel 16 (r >> 2) + 4 g + (r & 3)
        t_wr[(16 * (r >> 2) + (r & 3)) * V2_TT] = d2[r];
        t_wr[(32 + 16 * (r >> 2) + (r & 3)) * V2_TT] = h1[r];
      }
      wave_lds_fence();
      {
        float4 la[2], lb[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          la[m] = *reinterpret_cast<const float4*>(t_rd + (16 * m) * V2_TT);
          lb[m] = *reinterpret_cast<const float4*>(t_rd + (32 + 16 * m) * V2_TT);
          db2acc[m] += (la[m].x + la[m].y) + (la[m].z + la[m].w);  // db2 rides on the transposed operands
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) {  // dW2[out][in] += d2[out][k] h1[in][k]
            accW2[m][n] = mfma16(la[m].x, lb[n].x, accW2[m][n]);
            accW2[m][n] = mfma16(la[m].y, lb[n].y, accW2[m][n]);
            accW2[m][n] = mfma16(la[m].z, lb[n].z, accW2[m][n]);
            accW2[m][n] = mfma16(la[m].w, lb[n].w, accW2[m][n]);
          }
      }
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 8; ++r) t_wr[(16 * (r >> 2) + (r & 3)) * V2_TT] = d1[r];
      f_wr[32 * V2_TT] = f2[0];
      f_wr[33 * V2_TT] = f2[1];
      wave_lds_fence();
      {
        float4 la[2];
        // B columns 0..7 = f, column 8 = ones: accW1[:, 8] accumulates db1 = sum_k d1[ch][k] in the spare MFMA lanes
        float4 lb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i16 < F) lb = *reinterpret_cast<const float4*>(t_rd + 32 * V2_TT);
        if (i16 == F) lb = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
        for (int m = 0; m < 2; ++m) la[m] = *reinterpret_cast<const float4*>(t_rd + (16 * m) * V2_TT);
#pragma unroll
        for (int m = 0; m < 2; ++m) {  // dW1[ch][feat] += d1[ch][k] f[feat][k]
          accW1[m] = mfma16(la[m].x, lb.x, accW1[m]);
          accW1[m] = mfma16(la[m].y, lb.y, accW1[m]);
          accW1[m] = mfma16(la[m].z, lb.z, accW1[m]);
          accW1[m] = mfma16(la[m].w, lb.w, accW1[m]);
        }
      }
      wave_lds_fence();
    }
    __builtin_amdgcn_sched_barrier(0);  // phase boundary: no operand of the next phase is fetched early
    SHINE_STAMP(5)  // weight grads

    // ================================================================ phase 6: feature-grad scatter (run-length)
    f_wr[V2_DF] = df2[0];
    f_wr[V2_DF + V2_DFP] = df2[1];
    wave_lds_fence();
    {
      constexpr int CH = 8;  // points per chunk (register budget)
      unsigned int mymiss = 0u;  // this lane's trash level: the points that miss level sc
#pragma unroll
      for (int s = 0; s < L; ++s)
        if (sc == s) mymiss = ~hitmask[s] & validmask;
#pragma unroll
      for (int ch = 0; ch < V2_TP / CH; ++ch) {
        float dfr[CH];
#pragma unroll
        for (int j = 0; j < CH / 4; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(sc_df + CH * ch + 4 * j);
          dfr[4 * j] = v.x;
          dfr[4 * j + 1] = v.y;
          dfr[4 * j + 2] = v.z;
          dfr[4 * j + 3] = v.w;
        }
        {  // trash rows: the plain sum of df over the misses (the 8 corner weights of a missed node sum to 1)
          const unsigned int mm = mymiss >> (CH * ch);
#pragma unroll
          for (int p2 = 0; p2 < CH; ++p2) {
            const unsigned int keep = 0u - ((mm >> p2) & 1u);
            trash_sum += __uint_as_float(__float_as_uint(dfr[p2]) & keep);
          }
        }
#pragma unroll
        for (int s = 0; s < L; ++s) {
          float* gbase = a.lv[s].grad;
          if (gbase) {
            float wr[CH];
            int idr[CH];
#pragma unroll
            for (int j = 0; j < CH / 4; ++j) {
              const float4 v = *reinterpret_cast<const float4*>(sc_w + (s * 8) * V2_WP + CH * ch + 4 * j);
              wr[4 * j] = v.x;
              wr[4 * j + 1] = v.y;
              wr[4 * j + 2] = v.z;
              wr[4 * j + 3] = v.w;
              const int4 u = *reinterpret_cast<const int4*>(sc_ids + (s * 8) * V2_WP + CH * ch + 4 * j);
              idr[4 * j] = u.x;
              idr[4 * j + 1] = u.y;
              idr[4 * j + 2] = u.z;
              idr[4 * j + 3] = u.w;
            }
            int rid = run_id[s], rhit = run_hit[s];
            float racc = run_acc[s];
            const unsigned int cm = (chgmask[s] & validmask) >> (CH * ch);
            const unsigned int hm = hitmask[s] >> (CH * ch);
#pragma unroll
            for (int p2 = 0; p2 < CH; ++p2) {
              if (cm & (1u << p2)) {  // a new node (or a run of misses) starts here: close the open run
                if (rhit) atomic_add_f32(gbase + (unsigned int)rid, racc);  // scalar branch (wave-uniform hit mask)
                racc = 0.f;
                rid = (idr[p2] << 3) | sq;  // float offset of this lane's (corner row, feature)
                rhit = (int)((hm >> p2) & 1u);
              }
              racc = fmaf(wr[p2], dfr[p2], racc);  // misses and padding lanes staged w = 0
            }
            run_id[s] = rid;
            run_hit[s] = rhit;
            run_acc[s] = racc;
          }
          __builtin_amdgcn_sched_barrier(0);  // one level's staged operands in registers at a time
        }
      }
    }
    wave_lds_fence();
    SHINE_STAMP(4)  // scatter
  }

  // ---- end of the wave's run: flush the open node runs
#pragma unroll
  for (int s = 0; s < L; ++s) {
    float* gbase = a.lv[s].grad;
    if (gbase && run_hit[s]) atomic_add_f32(gbase + (unsigned int)run_id[s], run_acc[s]);
  }
  __syncthreads();  // every wave is done with its staging region: it now holds the wave's partial vector
  float* wvec = s_wave[wv];
  if (sc < L) wvec[PART_TRASH + sc * 8 + sq] = trash_sum;
  if (a.decoder_grad_on) {
    const int jc = lane & 15, rr = lane >> 4;  // accumulator role: column jc, rows 4 rr + r
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * m + 4 * rr + r;
        wvec[MLP_W2 + row * H + jc] = accW2[m][0][r];
        wvec[MLP_W2 + row * H + 16 + jc] = accW2[m][1][r];
        if (jc <= F) wvec[jc < F ? MLP_W1 + row * F + jc : MLP_B1 + row] = accW1[m][r];  // column 8 of accW1 is db1
        const float w3v = row16_sum(dw3c[4 * m + r]);  // channel 16 m + 4 g + r over the 16 points of the DPP row
        if (pt == 0) wvec[MLP_W3 + 16 * m + 4 * g + r] = w3v;
      }
#pragma unroll
    for (int m = 0; m < 2; ++m) {  // channel 16 m + (lane & 15), the four point groups kk
      float v = db2acc[m];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 16) wvec[MLP_B2 + 16 * m + lane] = v;
    }
    const float b3v = wave_sum(db3);
    if (lane == 0) wvec[MLP_B3] = b3v;
  }
  {
    const double ls = wave_sum_d((double)loss_acc), cs = wave_sum_d((double)cnt_acc);
    if (lane == 0) {
      atomicAdd(&s_loss[0], ls);
      atomicAdd(&s_loss[1], cs);
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
    dl[2] = 0.0;
  }
#undef SHINE_STAMP
}

// D[16x16] = A[16x4] . B[4x16] through one v_mfma_f32_16x16x4_f32: pins the operand / accumulator lane maps
__global__ void k_selftest_mfma16(const float* A, const float* B, float* D) {
  const int lane = threadIdx.x;
  f32x4 c = zero4();
  c = mfma16(A[(lane & 15) * 4 + (lane >> 4)], B[(lane >> 4) * 16 + (lane & 15)], c);
#pragma unroll
  for (int r = 0; r < 4; ++r) D[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];
}

V2Geometry v2_geometry(long long n) {
  V2Geometry g;
  long long tiles = (n + V2_TP - 1) / V2_TP;
  if (tiles < 1) tiles = 1;
  g.wg_waves = tiles < 2048 ? 4 : V2_BIG;  // small batches: 4-wave workgroups spread over more CUs
  const long long max_waves = 256 * V2_BIG;  // resident waves per CU
  g.waves = tiles < max_waves ? tiles : max_waves;
  const long long per = (n + g.waves - 1) / g.waves;
  g.chunk = ((per + V2_TP - 1) / V2_TP) * V2_TP;
  g.waves = (n + g.chunk - 1) / g.chunk;
  if (g.waves < 1) g.waves = 1;
  g.blocks = (g.waves + g.wg_waves - 1) / g.wg_waves;
  return g;
}

long long v2_lds_bytes(int wg_waves) {
  return (long long)sizeof(float) * (V2_OPTOTAL + 100 + (long long)wg_waves * V2_WAVE_FLOATS) + 4 * sizeof(double);
}

template <int L>
static void launch_v2(const V1Args& a, const V2Geometry& g, hipStream_t st) {
  const dim3 grid((unsigned)g.blocks);
  if (a.prof) {
    if (g.wg_waves == V2_BIG) hipLaunchKernelGGL((k_step_v2<L, V2_BIG, true>), grid, dim3(V2_BIG * 64), 0, st, a);
    else hipLaunchKernelGGL((k_step_v2<L, 4, true>), grid, dim3(256), 0, st, a);
    return;
  }
  if (g.wg_waves == V2_BIG) hipLaunchKernelGGL((k_step_v2<L, V2_BIG, false>), grid, dim3(V2_BIG * 64), 0, st, a);
  else hipLaunchKernelGGL((k_step_v2<L, 4, false>), grid, dim3(256), 0, st, a);
}

}  // namespace shine

using namespace shine;

extern "C" int shine_selftest_mfma16(const float* a, const float* b, float* d, void* stream) {
  if (!a || !b || !d) return set_error(SHINE_E_INVALID, "shine_selftest_mfma16: null argument");
  hipLaunchKernelGGL(k_selftest_mfma16, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, d);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

// same contract as shine_train_step_v1 (shine_step_v1.hip); BCE only, needs the partial-sum workspace
extern "C" int shine_train_step_v2(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                                   const float* sdf_label, const float* weight, const int32_t* perm,
                                   const int32_t* slots, const int64_t* n_surf, int64_t n, const float* const* feats,
                                   const int64_t* rows, const float* const* mlp, float* pred_out, float* grad_x_out,
                                   float* const* grad_feats, float* const* grad_mlp, double* loss_parts,
                                   unsigned char* const* touched, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  if (cfg->eikonal_on) return set_error(SHINE_E_INVALID, "shine_train_step_v2: BCE only");
  V1Args a = {};
  int rc = fill_step_args(&a, t, cfg, coord, sdf_label, weight, perm, slots, n_surf, n, feats, rows, mlp, pred_out,
                          grad_x_out, grad_feats, grad_mlp, loss_parts, touched);
  if (rc != SHINE_OK) return rc;
  if (n == 0) return SHINE_OK;
  const V2Geometry g = v2_geometry(n);
  a.chunk = g.chunk;
  a.prof = g_prof_buffer;
  const size_t need = (size_t)g.blocks * PART_STRIDE * sizeof(float);
  if (!workspace || workspace_bytes < need)
    return set_error(SHINE_E_INVALID, "shine_train_step_v2: workspace too small (shine_train_step_workspace_bytes)");
  a.partials = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  if (touched) {
    hipLaunchKernelGGL(k_mark_touched, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  switch (cfg->n_levels) {
    case 1: launch_v2<1>(a, g, st); break;
    case 2: launch_v2<2>(a, g, st); break;
    case 3: launch_v2<3>(a, g, st); break;
    default: launch_v2<4>(a, g, st); break;
  }
  SHINE_HIP_CHECK(hipGetLastError());
  if (!(a.ablate & 32)) {  // (ablate bit 32: measurement only — time the dominant kernel by itself)
    hipLaunchKernelGGL(k_reduce_partials, dim3((PART_FLOATS + 63) / 64), dim3(1024), 0, st, a, (int)g.blocks);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  return SHINE_OK;
}
