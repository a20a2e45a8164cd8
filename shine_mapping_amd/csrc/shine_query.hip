// shine_query.hip — forward-only SDF query for meshing: Mesher.query_points (utils/mesher.py:33-108), i.e.
// FeatureOctree.query_feature(coord, faster=True) (model/feature_octree.py:237-244 with get_indices_fast :267-286,
// same result as get_indices) + Decoder.sdf (model/decoder.py:49-63), the sign flip of :69/:92 and the
// marching-cubes mask of :78-86/:99-104 — one launch, nothing but the two outputs leaves the chip.
//
// Grid-structured queries are perfectly coherent (neighbouring lanes share voxels), so the gathers hit L1/L2 and
// the kernel is bound by the decoder arithmetic: lane = point, the 1377 decoder weights are read through the constant
// address space (scalar loads, SGPR operands: no VGPR or LDS read per weight), h1 stays in registers, h2 is consumed as
// it is produced.
// Bytes per query: 12 in, 5 out.
#include "shine_internal.hpp"

namespace shine {

struct QueryArgs {
  LevelSet ls;
  const float* coord;
  const float* mlp[6];
  float* sdf_out;          // [n] = -Decoder.sdf (mesher.py:69) or null
  unsigned char* mask_out; // [n] 1 iff all 8 corner ids of the check level are >= 0, or null
  long long n;
  int check_slot;          // top-down slot of the level the mask tests
  float sign;              // -1 (mesher) or +1 (plain Decoder.sdf)
};

template <int L, bool POLY>
__global__ __launch_bounds__(256) void k_query_points(QueryArgs a) {
  // decoder weights: wave-uniform, read-only -> scalar loads into SGPRs that the FMAs take as operands (device.hpp)
  cfloat* W1 = nullptr, *B1 = nullptr, *W2 = nullptr, *B2 = nullptr, *W3 = nullptr, *B3 = nullptr;
  if (a.sdf_out) {  // wave-uniform
    W1 = uniform_ro(a.mlp[0]), B1 = uniform_ro(a.mlp[1]), W2 = uniform_ro(a.mlp[2]), B2 = uniform_ro(a.mlp[3]);
    W3 = uniform_ro(a.mlp[4]), B3 = uniform_ro(a.mlp[5]);
  }
  const int rows = opaque(H);

  const long long stride = (long long)gridDim.x * 256;
  for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < a.n; p += stride) {
    const float x0 = a.coord[3 * p], x1 = a.coord[3 * p + 1], x2 = a.coord[3 * p + 2];
    float f[F];
#pragma unroll
    for (int k = 0; k < F; ++k) f[k] = 0.f;
    bool mask = false;
    // all probes first (independent loads), then the gathers
    int slot[L];
#pragma unroll
    for (int s = 0; s < L; ++s) {
      const LevelDev& Lv = a.ls.lv[s];
      const unsigned long long key = morton3(quantize(x0, Lv.res), quantize(x1, Lv.res), quantize(x2, Lv.res));
      slot[s] = probe(Lv, key);
    }
#pragma unroll
    for (int s = 0; s < L; ++s) {
      const LevelDev& Lv = a.ls.lv[s];
      const bool hit = slot[s] >= 0;
      if (s == a.check_slot) mask = hit;
      const unsigned int sl = hit ? (unsigned int)slot[s] : 0u;
      const int4 i0 = Lv.vals[2u * sl], i1 = Lv.vals[2u * sl + 1u];
      const int ids[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
      const Axis X = axis_weight<POLY>(x0, Lv.res, Lv.dres), Y = axis_weight<POLY>(x1, Lv.res, Lv.dres),
                 Z = axis_weight<POLY>(x2, Lv.res, Lv.dres);
      float w[8];
      corner_weights(X.t, Y.t, Z.t, w);
      float4 r0[8], r1[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const unsigned int off = (hit ? (unsigned int)ids[c] : 0u) * (unsigned int)F;
        r0[c] = *reinterpret_cast<const float4*>(Lv.feat + off);
        r1[c] = *reinterpret_cast<const float4*>(Lv.feat + off + 4u);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float wz = hit ? w[c] : 0.f;
        f[0] += wz * r0[c].x;
        f[1] += wz * r0[c].y;
        f[2] += wz * r0[c].z;
        f[3] += wz * r0[c].w;
        f[4] += wz * r1[c].x;
        f[5] += wz * r1[c].y;
        f[6] += wz * r1[c].z;
        f[7] += wz * r1[c].w;
      }
    }
    if (a.sdf_out) {
      float h1[H];
      cfloat *const W1i = relaunder(W1), *const B1i = relaunder(B1);  // keep the loads inside this iteration
#pragma unroll
      for (int j = 0; j < H; ++j) {
        float z = B1i[j];
#pragma unroll
        for (int q = 0; q < F; ++q) z = fmaf(W1i[j * F + q], f[q], z);
        h1[j] = fmaxf(z, 0.f);
      }
      float y = B3[0];
      // layer 2 + 3 as a ROLLED loop over the 32 weight rows (two s_load_dwordx16 each): h2 is consumed as produced
#pragma clang loop vectorize(disable) interleave(disable) unroll_count(2)
      for (int j = 0; j < rows; ++j) {
        float z = B2[j];
#pragma unroll
        for (int k = 0; k < H; ++k) z = fmaf(W2[j * H + k], h1[k], z);
        y = fmaf(W3[j], fmaxf(z, 0.f), y);
      }
      a.sdf_out[p] = a.sign * y;
    }
    if (a.mask_out) a.mask_out[p] = mask ? 1 : 0;
  }
}

template <int L>
static void launch_query(const QueryArgs& a, bool poly, hipStream_t st) {
  long long blocks = (a.n + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 workgroups per CU
  if (poly)
    hipLaunchKernelGGL((k_query_points<L, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((k_query_points<L, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
}

}  // namespace shine

using namespace shine;

extern "C" int shine_query_points(const shine_tables* t, const shine_step_config* cfg, const float* coord, int64_t n,
                                  const float* const* feats, const int64_t* rows, const float* const* mlp,
                                  int32_t check_level, int32_t negate, float* sdf_out, uint8_t* mask_out,
                                  void* stream) {
  if (n < 0 || !feats || !rows || (n > 0 && !coord)) return set_error(SHINE_E_INVALID, "shine_query_points: null argument");
  if (sdf_out && !mlp) return set_error(SHINE_E_INVALID, "shine_query_points: sdf output needs the decoder");
  QueryArgs a = {};
  int rc = make_level_set(t, cfg, feats, rows, nullptr, &a.ls);
  if (rc != SHINE_OK) return rc;
  const int L = cfg->n_levels;
  if (check_level < 0 || check_level >= L) return set_error(SHINE_E_INVALID, "shine_query_points: check_level out of range");
  for (int s = 0; s < L; ++s) {
    if (!feats[s]) return set_error(SHINE_E_INVALID, "shine_query_points: null feature level");
    if (rows[s] >= (1ll << 29)) return set_error(SHINE_E_INVALID, "shine_query_points: level exceeds 2^29 rows");
  }
  if (sdf_out)
    for (int k = 0; k < 6; ++k) {
      if (!mlp[k]) return set_error(SHINE_E_INVALID, "shine_query_points: null decoder parameter");
      a.mlp[k] = mlp[k];
    }
  if (n == 0 || (!sdf_out && !mask_out)) return SHINE_OK;
  a.coord = coord;
  a.sdf_out = sdf_out;
  a.mask_out = mask_out;
  a.n = n;
  a.check_slot = L - 1 - check_level;  // hierarchical_indices is bottom-up (mesher.py:78), slots are top-down
  a.sign = negate ? -1.f : 1.f;
  const bool poly = cfg->poly_int_on != 0;
  hipStream_t st = (hipStream_t)stream;
  switch (L) {
    case 1: launch_query<1>(a, poly, st); break;
    case 2: launch_query<2>(a, poly, st); break;
    case 3: launch_query<3>(a, poly, st); break;
    case 4: launch_query<4>(a, poly, st); break;
    default: return set_error(SHINE_E_INVALID, "shine_query_points: more than 4 featured levels");
  }
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}
