// shine_tile16.hpp — the 16-point-tile geometry of the fused step (shine_step_v3.hip: one wave does every phase of a tile):
// the staging layout, the pre-permuted decoder operand image, and the cross-lane helpers (v_permlane{16,32}_swap
// reduce-scatter, DPP row moves).
#pragma once
#include "shine_step_common.hpp"

namespace shine {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// packed fp32 (gfx950: v_pk_fma_f32 / v_pk_mul_f32 do two fp32 operations per lane; a splat operand costs nothing, op_sel
// broadcasts one half of the source pair)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return (f32x2){v, v}; }

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

// LDS of one workgroup of the fused step: the decoder operand image, biases / w3 / b3, the loss accumulators and one staging
// region per wave (which ends its life as the wave's partial vector).
// CACHE (the FAR build, shine_step_body.hpp): per wave and level a 4 x 4 x 4 lattice of corner rows — 64 tags (row ids, -1 =
// empty) and 64 x 8 partial gradient sums — in which closed node runs are merged before they go to memory as atomics.
constexpr int V3_CSLOTS = 64;                              // slot = (cx & 3) << 4 | (cy & 3) << 2 | (cz & 3)
constexpr int V3_CLEVEL = V3_CSLOTS + V3_CSLOTS * F;       // floats per level: tags, then values [slot][feature]
constexpr int V3_CACHE_FLOATS = LCAP * V3_CLEVEL;          // 2304 floats = 9216 B per wave
constexpr int V3_ROW_BITS = 25;                            // FAR: a staged corner id carries its lattice slot above bit 25
template <int WAVES, bool CACHE = false>
struct StepShared {
  alignas(16) float opA[V3_OPTOTAL];
  float bias[100];  // (directly behind opA: the operand image is copied over both)
  double loss[4];
  float wave[WAVES][V3_WAVE_FLOATS];
  float cache[CACHE ? WAVES : 1][CACHE ? V3_CACHE_FLOATS : 4];
};

// Per-workgroup setup: the decoder's A operands pre-permuted into 16x16x4 lane order (lane l: row i = l & 15, k-group
// kg = l >> 4) and the bias / w3 vectors.  Branch-free source select, fully unrolled: the loads of a thread are all in flight
// together (one L2 round trip).  s_opA: V3_OPTOTAL floats, s_bias: 100 floats.  The values are loaded into registers by
// decoder_operands_issue (so that the caller can put other loads in flight behind them) and stored by decoder_operands_store.
template <int NT>
struct OperandRegs {
  float v[(V3_OPTOTAL + NT - 1) / NT];
  float b[3];
  float b3;
};

template <int NT>
__device__ __forceinline__ void decoder_operands_issue(const V1Args& a, OperandRegs<NT>& R, int tid) {
#pragma unroll
  for (int it = 0; it < (V3_OPTOTAL + NT - 1) / NT; ++it) {
    const int idx = it * NT + tid;
    R.v[it] = 0.f;
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
      R.v[it] = zero ? 0.f : v;
    }
  }
  R.b[0] = R.b[1] = R.b[2] = R.b3 = 0.f;
  if (tid < 32) {
    R.b[0] = a.mlp[1][tid];
    R.b[1] = a.mlp[3][tid];
    R.b[2] = a.mlp[4][tid];
  }
  if (tid == 0) R.b3 = a.mlp[5][0];
}

template <int NT>
__device__ __forceinline__ void decoder_operands_store(const OperandRegs<NT>& R, float* s_opA, float* s_bias, int tid) {
#pragma unroll
  for (int it = 0; it < (V3_OPTOTAL + NT - 1) / NT; ++it) {
    const int idx = it * NT + tid;
    if (idx < V3_OPTOTAL) s_opA[idx] = R.v[it];
  }
  if (tid < 32) {
    s_bias[tid] = R.b[0];
    s_bias[32 + tid] = R.b[1];
    s_bias[64 + tid] = R.b[2];
  }
  if (tid == 0) s_bias[96] = R.b3;
}

// The inverse of decoder_operands_issue's map, for the launch that UPDATES the decoder (k_finish): where element e of a decoder
// tensor lives in the operand image (opA then the 100 bias floats).  part_off = the tensor's offset in the flat parameter order
// (MLP_W1, MLP_B1, ...).  Up to two places (a weight matrix appears as itself and transposed); -1: none.
__device__ __forceinline__ void operand_image_slots(int part_off, int e, int& i0, int& i1) {
  i0 = i1 = -1;
  if (part_off == MLP_W1) {  // [H][F]: row R, column C
    const int R = e >> 3, C = e & 7;
    i0 = ((R >> 4) * 2 + (C & 1)) * 64 + (C >> 1) * 16 + (R & 15);
    i1 = (36 + (R >> 4) * 4 + (R & 3)) * 64 + ((R >> 2) & 3) * 16 + 4 * (C >> 1) + (C & 1);
  } else if (part_off == MLP_W2) {  // [H][H]: row r, column c
    const int r = e >> 5, c = e & 31;
    i0 = (4 + (r >> 4) * 8 + (c >> 4) * 4 + (c & 3)) * 64 + ((c >> 2) & 3) * 16 + (r & 15);
    i1 = (20 + (c >> 4) * 8 + (r >> 4) * 4 + (r & 3)) * 64 + ((r >> 2) & 3) * 16 + (c & 15);
  } else if (part_off == MLP_B1) {
    i0 = V3_OPTOTAL + e;
  } else if (part_off == MLP_B2) {
    i0 = V3_OPTOTAL + 32 + e;
  } else if (part_off == MLP_W3) {
    i0 = V3_OPTOTAL + 64 + e;
  } else if (part_off == MLP_B3) {
    i0 = V3_OPTOTAL + 96;
  }
}
constexpr int V3_IMAGE_FLOATS = V3_OPTOTAL + 100;  // 2916 = 729 float4
static_assert(V3_IMAGE_FLOATS % 4 == 0, "the image is copied as float4");

}  // namespace shine
