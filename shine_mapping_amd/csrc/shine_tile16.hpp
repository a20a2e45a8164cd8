// shine_tile16.hpp — what the 16-point-tile fused-step kernels share (shine_step_v3.hip: one wave does every phase of a
// tile; shine_step_v5.hip: role-specialised waves around per-SIMD LDS rings): the staging geometry, the pre-permuted
// decoder operand image, and the cross-lane helpers (v_permlane{16,32}_swap reduce-scatter, DPP row moves).
#pragma once
#include "shine_step_common.hpp"

namespace shine {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int V3_TP = 16;                      // points per tile
constexpr int V3_WP = 16;                      // pitch of the [corner][point] staging rows
constexpr int V3_TT = 20;                      // transpose tile pitch (floats, 16-B aligned rows)
constexpr int V3_DFP = 20;                     // pitch of the [feature][point] rows
constexpr int V3_IDS = LCAP * 8 * V3_WP;       // ids [LCAP][8][16] int32
constexpr int V3_W = LCAP * 8 * V3_WP;         // w   [LCAP][8][16]
constexpr int V3_R2 = 2 * 32 * V3_TT;          // two transpose tiles [32][20]; second life: df
constexpr int V3_DF = 0, V3_DL = 8 * V3_DFP;    // df / J rows [8][20], then delta[16] (eikonal build)
constexpr int V3_WAVE_FLOATS = V3_IDS + V3_W + V3_R2;  // 2304 floats = 9216 B per wave
constexpr int V3_OPA1 = 0, V3_OPA2 = 4 * 64, V3_OPA2T = 20 * 64, V3_OPA1T = 36 * 64, V3_OPTOTAL = 44 * 64;
static_assert(V3_DFP == V3_TT, "f_wr addresses both the transpose rows and the df rows");
static_assert(PART_STRIDE <= V3_WAVE_FLOATS, "each wave's partial vector aliases its staging region at the end");

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 zero4() {
  f32x4 z;
  z[0] = z[1] = z[2] = z[3] = 0.f;
  return z;
}

// sigmoid on the hardware transcendental units: 1 / (1 + 2^(-x log2 e)); exp2 overflow -> rcp(inf) = 0, as it should
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * x));
}

// x + (the value lane ^ 32 holds in y)  for lanes < 32,   y + (the value lane ^ 32 holds in x)  for lanes >= 32:
// v_permlane32_swap exchanges x[32..63] with y[0..31], after which both registers hold one own and one partner value.
__device__ __forceinline__ float xsum32(float x, float y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// the same across lane ^ 16 (v_permlane16_swap: x rows 1, 3 <-> y rows 0, 2)
__device__ __forceinline__ float xsum16(float x, float y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// lane 15 of each 16-lane row to every lane of the row (DPP row_newbcast:15), lane i-1 of the row to lane i (row_shr:1)
__device__ __forceinline__ int row_last(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x15F, 0xF, 0xF, false); }
__device__ __forceinline__ int row_prev(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x111, 0xF, 0xF, false); }

// Per-workgroup setup shared by the 16-point-tile kernels: the decoder's A operands pre-permuted into 16x16x4 lane order
// (lane l: row i = l & 15, k-group kg = l >> 4) and the bias / w3 vectors.  Branch-free source select, fully unrolled: the
// loads of a thread are all in flight together (one L2 round trip).  s_opA: V3_OPTOTAL floats, s_bias: 100 floats.
template <int NT>
__device__ __forceinline__ void load_decoder_operands(const V1Args& a, float* s_opA, float* s_bias, int tid) {
#pragma unroll
  for (int it = 0; it < (V3_OPTOTAL + NT - 1) / NT; ++it) {
    const int idx = it * NT + tid;
    if (idx < V3_OPTOTAL) {
      const int t = idx >> 6, l = idx & 63, i = l & 15, kg = l >> 4;
      const float* src;
      bool zero = false;
      if (t < 4) {  // W1: M-block mb = t >> 1, k-step tt = t & 1 contracts over features 2 kg + tt
        src = a.mlp[0] + (16 * (t >> 1) + i) * F + 2 * kg + (t & 1);
      } else if (t < 20) {  // W2: mb, k-step (m', r) contracts over channels 16 m' + 4 kg + r
        const int u = t - 4, mb = u >> 3, ks = u & 7;
        src = a.mlp[2] + (16 * mb + i) * H + 16 * (ks >> 2) + 4 * kg + (ks & 3);
      } else if (t < 36) {  // W2^T
        const int u = t - 20, mb = u >> 3, ks = u & 7;
        src = a.mlp[2] + (16 * (ks >> 2) + 4 * kg + (ks & 3)) * H + 16 * mb + i;
      } else {  // W1^T, output rows permuted: row 4 g' + r' = feature 2 g' + r' for r' < 2, zero otherwise
        const int ks = t - 36, gp = i >> 2, rp = i & 3;
        src = a.mlp[0] + (16 * (ks >> 2) + 4 * kg + (ks & 3)) * F + 2 * gp + (rp & 1);
        zero = rp >= 2;
      }
      const float v = *src;
      s_opA[idx] = zero ? 0.f : v;
    }
  }
  if (tid < 32) {
    s_bias[tid] = a.mlp[1][tid];
    s_bias[32 + tid] = a.mlp[3][tid];
    s_bias[64 + tid] = a.mlp[4][tid];
  }
  if (tid == 0) s_bias[96] = a.mlp[5][0];
}

// ---- feature-grad scatter of ONE level of ONE 16-point tile, lane = (corner, feature): prefix-sum form of the run-length walk.
// The serial walk (test a run-start bit, branch, v_fmac — 16 times per level) is a chain of scalar compare -> taken branch ->
// dependent VALU: ~60 cycles per point in the role-specialised kernel's scatter wave (profiles/r03_ab_experiments.txt block 3).
// Here the 16 products are accumulated unconditionally into 16 REGISTERS P[p] = sum_{p' <= p} w[p'] df[p'] (16 v_fma, no
// scalar work), and only the node runs that END inside the tile are visited — a wave-uniform loop over the set bits of the
// run-start mask — taking  sum(run) = P[end - 1] - P[start - 1]  with a uniform dynamic register index (s_set_gpr_idx: no
// LDS, no scratch).  A tile without a run start (the usual case at the coarse levels) is 16 fma and one add.
// Rounding: a run's sum is a difference of two prefix sums of at most 16 terms — ~1e-6 of the largest term, against the
// 1e-4-of-max-abs contract; the run's first partial (carried in from earlier tiles) is added exactly as before.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <bool ATOMICS>
__device__ __forceinline__ void scatter_level_prefix(const f32x16& wr, const i32x16& idr, const f32x16& dfr, unsigned int cm,
                                                     unsigned int hm, int sq, float* gbase, unsigned char* tb, int& rid,
                                                     int& rhit, float& racc) {
  f32x16 P;
  P[0] = wr[0] * dfr[0];
#pragma unroll
  for (int p = 1; p < V3_TP; ++p) P[p] = fmaf(wr[p], dfr[p], P[p - 1]);  // misses and padding lanes staged w = 0
  if (cm == 0u) {  // the open run covers the whole tile
    racc += P[V3_TP - 1];
    return;
  }
  unsigned int m = cm;
  float prev = 0.f;
  do {
    const int st = __builtin_ctz(m);  // wave-uniform: a new node (or a run of misses) starts at point st
    m &= m - 1u;
    // (the index is forced into an SGPR: behind a VALU-derived select the compiler emits a waterfall loop around the move)
    const int i0 = __builtin_amdgcn_readfirstlane(st > 0 ? st - 1 : 0);
    float pv = P[i0];
    if (st == 0) pv = 0.f;
    if (rhit && ATOMICS) atomic_add_f32(gbase + (unsigned int)rid, racc + (pv - prev));  // close the open run
    racc = 0.f;
    prev = pv;
    const int id = idr[st];
    rid = (id << 3) | sq;  // float offset of this lane's (corner row, feature)
    rhit = (int)((hm >> st) & 1u);
    // touched-row flags (unique(hierarchical_indices) without -1): set at the run start of every hit node, one lane per corner
    if (tb && rhit && sq == 0) tb[id] = 1;
  } while (m);
  racc = P[V3_TP - 1] - prev;  // the partial of the run that stays open
}

}  // namespace shine
