// shine_device.hpp — device-side building blocks shared by every gfx950 kernel of the SHINE hot path.
//
// Coordinate arithmetic is the one place where the result must be BIT-identical to the reference's
// fp32 torch ops: the SDF field is discontinuous at allocated/empty voxel borders, so a voxel id that
// flips by one ulp produces O(1) errors.  Those expressions are written with __f*_rn intrinsics so
// that no fma contraction or re-association can touch them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/shine_hip.h"

namespace shine {

constexpr int F = SHINE_FEATURE_DIM;  // 8
constexpr int H = SHINE_HIDDEN_DIM;   // 32
constexpr int MLP_W1 = 0;             // [H][F]
constexpr int MLP_B1 = MLP_W1 + H * F;
constexpr int MLP_W2 = MLP_B1 + H;    // [H][H]
constexpr int MLP_B2 = MLP_W2 + H * H;
constexpr int MLP_W3 = MLP_B2 + H;    // [H]
constexpr int MLP_B3 = MLP_W3 + H;
static_assert(MLP_B3 + 1 == SHINE_MLP_PARAMS, "packed decoder layout");

constexpr unsigned long long EMPTY_KEY = ~0ull;

// One featured level as the kernels see it (slot order top-down, see shine_hip.h).
struct LevelDev {
  const unsigned long long* keys;  // [cap] node Morton code or EMPTY_KEY
  const int4* vals;                // [cap][2] = 8 int32 corner ids
  const float* feat;               // [rows+1][F]
  float* grad;                     // [rows+1][F] or nullptr
  long long rows;                  // trash row index
  unsigned int shift;              // 64 - log2(cap)
  unsigned int mask;               // cap - 1
  float res;                       // 2^level
  float dres;                      // res / 2  (= d u / d x)
};

struct LevelSet {
  LevelDev lv[SHINE_MAX_LEVELS];
};

// ---- kaolin.ops.spc.points_to_morton (call site model/feature_octree.py:204): x is the MSB of each triplet
__device__ __forceinline__ unsigned long long spread3(unsigned int v) {
  unsigned long long x = v & 0xFFFFu;
  x = (x | (x << 32)) & 0x001F00000000FFFFull;
  x = (x | (x << 16)) & 0x001F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}
__device__ __forceinline__ unsigned long long morton3(unsigned int x, unsigned int y, unsigned int z) {
  return (spread3(x) << 2) | (spread3(y) << 1) | spread3(z);
}

// ---- kaolin.ops.spc.quantize_points (call site model/feature_octree.py:203):
//      floor(clamp(res * (x + 1) / 2, 0, res - 1))  in fp32, exactly.
__device__ __forceinline__ unsigned int quantize(float x, float res) {
  float s = __fmul_rn(__fmul_rn(res, __fadd_rn(x, 1.0f)), 0.5f);
  s = fminf(fmaxf(s, 0.0f), __fsub_rn(res, 1.0f));
  return (unsigned int)floorf(s);
}

__device__ __forceinline__ unsigned int hash_slot(unsigned long long key, unsigned int shift) {
  return (unsigned int)((key * 0x9E3779B97F4A7C15ull) >> shift);
}

// Open-addressing probe. Returns the slot holding `key` or -1 (python dict.get(key, [-1]*8), :209).
__device__ __forceinline__ int probe(const LevelDev& L, unsigned long long key) {
  unsigned int slot = hash_slot(key, L.shift);
  for (unsigned int n = 0; n <= L.mask; ++n) {
    unsigned long long k = L.keys[slot];
    if (k == key) return (int)slot;
    if (k == EMPTY_KEY) return -1;
    slot = (slot + 1) & L.mask;
  }
  return -1;
}

// ---- FeatureOctree.interpolat (model/feature_octree.py:172-196), one axis:
//      u = res*(x*0.5+0.5); d = frac(u); t = 3d^2 - 2d^3 (or d); dt = dt/dx.
struct Axis {
  float t, dt;
};
template <bool POLY>
__device__ __forceinline__ Axis axis_weight(float x, float res, float dres) {
  float u = __fmul_rn(res, __fadd_rn(__fmul_rn(x, 0.5f), 0.5f));
  float d = __fsub_rn(u, truncf(u));
  Axis a;
  if (POLY) {
    float d2 = __fmul_rn(d, d);
    a.t = __fsub_rn(__fmul_rn(3.0f, d2), __fmul_rn(2.0f, __fmul_rn(d2, d)));
    a.dt = 6.0f * (d - d2) * dres;
  } else {
    a.t = d;
    a.dt = dres;
  }
  return a;
}

// corner c = 4*cx + 2*cy + cz; weight = phi_x * phi_y * phi_z in the reference's association (:186-193)
__device__ __forceinline__ void corner_weights(float tx, float ty, float tz, float w[8]) {
  float ux = __fsub_rn(1.0f, tx), uy = __fsub_rn(1.0f, ty), uz = __fsub_rn(1.0f, tz);
  float a00 = __fmul_rn(ux, uy), a01 = __fmul_rn(ux, ty), a10 = __fmul_rn(tx, uy), a11 = __fmul_rn(tx, ty);
  w[0] = __fmul_rn(a00, uz);
  w[1] = __fmul_rn(a00, tz);
  w[2] = __fmul_rn(a01, uz);
  w[3] = __fmul_rn(a01, tz);
  w[4] = __fmul_rn(a10, uz);
  w[5] = __fmul_rn(a10, tz);
  w[6] = __fmul_rn(a11, uz);
  w[7] = __fmul_rn(a11, tz);
}

// d w_c / d x_a for a = 0,1,2
__device__ __forceinline__ void corner_weight_grads(const Axis& X, const Axis& Y, const Axis& Z, float dw[8][3]) {
  float px[2] = {1.0f - X.t, X.t}, py[2] = {1.0f - Y.t, Y.t}, pz[2] = {1.0f - Z.t, Z.t};
  float gx[2] = {-X.dt, X.dt}, gy[2] = {-Y.dt, Y.dt}, gz[2] = {-Z.dt, Z.dt};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    int cx = (c >> 2) & 1, cy = (c >> 1) & 1, cz = c & 1;
    dw[c][0] = gx[cx] * py[cy] * pz[cz];
    dw[c][1] = px[cx] * gy[cy] * pz[cz];
    dw[c][2] = px[cx] * py[cy] * gz[cz];
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// wave-uniform read-only data (decoder weights) through the constant address space: uniform-address loads from it are
// scalar (s_load_dwordx*), and VALU instructions take the SGPR operand directly.  Only for memory no kernel of the same
// launch writes.
typedef const float __attribute__((address_space(4))) cfloat;
__device__ __forceinline__ cfloat* uniform_ro(const float* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (cfloat*)p;
#pragma clang diagnostic pop
}

// wave-local LDS hand-off: DS ops of one wave execute in order; this only stops the compiler reordering them
// (a workgroup-scope __builtin_amdgcn_fence would also drain vmcnt, i.e. wait for every gather/atomic in flight)
__device__ __forceinline__ void wave_lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Hide a uniform value / pointer from the optimiser (no instruction).  opaque(H) as a trip count keeps a loop over weight
// rows ROLLED (an `unroll 2` pragma alone is followed by a full unroll of the remaining 16 trips); relaunder(p) once per
// loop iteration keeps loads through p inside that iteration (they are loop-invariant, and 1377 hoisted weights is
// ~1300 spilled SGPRs).
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+s"(v));
  return v;
}
__device__ __forceinline__ cfloat* relaunder(cfloat* p) {
  asm volatile("" : "+s"(p));
  return p;
}

// hardware fp32 atomic add (global_atomic_add_f32), no CAS loop
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ---- lane-per-point kernels (shine_step_v0.hip, shine_mlp.hip): decoder weight grads contract over POINTS; the
//      per-point vectors of one wave are staged in LDS as [64 points][ST] and every lane owns a few output entries.
constexpr int ST = 75;  // staging row stride in floats (odd -> conflict-free column writes)

// contraction over the 64 staged points: acc[q] += left[p][j] * right[p][rb+q]
template <int NQ>
__device__ __forceinline__ void contract64(const float* st, int j, int rb, float (&acc)[NQ], float& acc_left,
                                           bool with_bias) {
  for (int p = 0; p < 64; ++p) {
    const float* row = st + p * ST;
    float l = row[j];
    if (with_bias) acc_left += l;
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = fmaf(l, row[rb + q], acc[q]);
  }
}

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace shine
