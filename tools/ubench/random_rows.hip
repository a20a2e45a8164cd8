// What the memory system delivers for the fused step's ACCESS STREAM alone (no decoder, no arithmetic): the roof the far build
// (tables beyond the Infinity Cache) is measured against.  Built as a shared library and driven by tools/ubench/random_rows.py on
// the corner ids of a real sorted batch:
//   rr_gather : lane = one (point, level) of the ordered stream, loads its 8 x 32-B corner rows (sixteen 16-B loads, like the
//               step's query phase) and keeps a sum
//   rr_scatter: one wave per node run, lane = (corner, feature): ONE 64-lane global_atomic_add_f32 covering 8 rows x 32 B
//               (the step's run-merged scatter)
//   rr_both   : even workgroups gather, odd workgroups scatter, in one launch (do the two streams add or overlap?)
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ab/librandom_rows.so tools/ubench/random_rows.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ float gather8(const float* feat, const int* ids) {
  const int4 a = *reinterpret_cast<const int4*>(ids), b = *reinterpret_cast<const int4*>(ids + 4);
  const int id[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  float4 r0[8], r1[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float* row = feat + (size_t)(unsigned int)(id[c] < 0 ? 0 : id[c]) * 8;
    r0[c] = *reinterpret_cast<const float4*>(row);
    r1[c] = *reinterpret_cast<const float4*>(row + 4);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += (r0[c].x + r0[c].y + r0[c].z + r0[c].w) + (r1[c].x + r1[c].y + r1[c].z + r1[c].w);
  return s;
}

// ids: [m][8] int32 per (point, level) entry in visiting order, lvl: [m] the entry's level; feat[l]: the level's table
struct Tabs {
  const float* feat[4];
  float* grad[4];
};

__global__ __launch_bounds__(256) void k_gather(Tabs t, const int* ids, const unsigned char* lvl, long long m, float* out,
                                                 int part, int parts) {
  const long long nb = gridDim.x / parts, b = blockIdx.x / parts;
  if ((int)(blockIdx.x % parts) != part) return;
  float acc = 0.f;
  // contiguous chunk per workgroup (the step gives a workgroup a contiguous piece of the ordered stream)
  const long long lo = b * m / nb, hi = (b + 1) * m / nb;
  for (long long e = lo + threadIdx.x; e < hi; e += 256) acc += gather8(t.feat[lvl[e]], ids + e * 8);
  if (acc == 123.456f) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_scatter(Tabs t, const int* ids, const unsigned char* lvl, long long m, int part,
                                                  int parts) {
  const long long nb = gridDim.x / parts, b = blockIdx.x / parts;
  if ((int)(blockIdx.x % parts) != part) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane >> 3, q = lane & 7;
  const long long lo = b * m / nb, hi = (b + 1) * m / nb;
  for (long long e = lo + wv; e < hi; e += 4) {
    const int id = ids[e * 8 + c];
    if (id >= 0) unsafeAtomicAdd(t.grad[lvl[e]] + (size_t)(unsigned int)id * 8 + q, 1.0f);
  }
}

// Round 6: what the SAME scatter stream costs when a row is updated without the L2's atomic units — a plain read-modify-write
// (valid in the step only for rows a wave owns exclusively; here the races are ignored: the question is the memory system's
// rate).  mode 0: DEPTH independent loads in flight per wave, then DEPTH adds + stores; mode 1: stores only (the upper bound).
template <int DEPTH, int MODE>
__global__ __launch_bounds__(256) void k_scatter_rmw(Tabs t, const int* ids, const unsigned char* lvl, long long m) {
  const long long nb = gridDim.x, b = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane >> 3, q = lane & 7;
  const long long lo = b * m / nb, hi = (b + 1) * m / nb;
  for (long long e0 = lo + wv * DEPTH; e0 < hi; e0 += 4 * DEPTH) {
    float* p[DEPTH];
    float v[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
      const long long e = e0 + k;
      const int id = e < hi ? ids[e * 8 + c] : -1;
      p[k] = id >= 0 ? t.grad[lvl[e < hi ? e : lo]] + (size_t)(unsigned int)id * 8 + q : nullptr;
    }
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < DEPTH; ++k) v[k] = p[k] ? *p[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < DEPTH; ++k)
      if (p[k]) *p[k] = MODE == 0 ? v[k] + 1.0f : 1.0f;
  }
}

__global__ __launch_bounds__(256) void k_both(Tabs t, const int* gids, const unsigned char* glvl, long long gm, const int* sids,
                                               const unsigned char* slvl, long long sm, float* out) {
  const long long nb = gridDim.x / 2, b = blockIdx.x / 2;
  if (blockIdx.x & 1) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c = lane >> 3, q = lane & 7;
    const long long lo = b * sm / nb, hi = (b + 1) * sm / nb;
    for (long long e = lo + wv; e < hi; e += 4) {
      const int id = sids[e * 8 + c];
      if (id >= 0) unsafeAtomicAdd(t.grad[slvl[e]] + (size_t)(unsigned int)id * 8 + q, 1.0f);
    }
  } else {
    float acc = 0.f;
    const long long lo = b * gm / nb, hi = (b + 1) * gm / nb;
    for (long long e = lo + threadIdx.x; e < hi; e += 256) acc += gather8(t.feat[glvl[e]], gids + e * 8);
    if (acc == 123.456f) out[0] = acc;
  }
}

extern "C" int rr_gather(const float* const* feat, const int* ids, const unsigned char* lvl, long long m, float* out, int blocks,
                         void* stream) {
  Tabs t = {};
  for (int l = 0; l < 4; ++l) t.feat[l] = feat[l];
  hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, ids, lvl, m, out, 0, 1);
  return (int)hipGetLastError();
}
extern "C" int rr_scatter(float* const* grad, const int* ids, const unsigned char* lvl, long long m, int blocks, void* stream) {
  Tabs t = {};
  for (int l = 0; l < 4; ++l) t.grad[l] = grad[l];
  hipLaunchKernelGGL(k_scatter, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, ids, lvl, m, 0, 1);
  return (int)hipGetLastError();
}
extern "C" int rr_both(const float* const* feat, float* const* grad, const int* gids, const unsigned char* glvl, long long gm,
                       const int* sids, const unsigned char* slvl, long long sm, float* out, int blocks, void* stream) {
  Tabs t = {};
  for (int l = 0; l < 4; ++l) t.feat[l] = feat[l], t.grad[l] = grad[l];
  hipLaunchKernelGGL(k_both, dim3(2 * blocks), dim3(256), 0, (hipStream_t)stream, t, gids, glvl, gm, sids, slvl, sm, out);
  return (int)hipGetLastError();
}

extern "C" int rr_scatter_rmw(float* const* grad, const int* ids, const unsigned char* lvl, long long m, int blocks, int depth,
                              int mode, void* stream) {
  Tabs t = {};
  for (int l = 0; l < 4; ++l) t.grad[l] = grad[l];
  hipStream_t st = (hipStream_t)stream;
  if (mode == 1) hipLaunchKernelGGL((k_scatter_rmw<8, 1>), dim3(blocks), dim3(256), 0, st, t, ids, lvl, m);
  else if (depth == 1) hipLaunchKernelGGL((k_scatter_rmw<1, 0>), dim3(blocks), dim3(256), 0, st, t, ids, lvl, m);
  else if (depth == 4) hipLaunchKernelGGL((k_scatter_rmw<4, 0>), dim3(blocks), dim3(256), 0, st, t, ids, lvl, m);
  else hipLaunchKernelGGL((k_scatter_rmw<8, 0>), dim3(blocks), dim3(256), 0, st, t, ids, lvl, m);
  return (int)hipGetLastError();
}
