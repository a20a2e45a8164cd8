// shine_sort.hip — Morton ordering of a point batch (new step right after LiDARDataset.get_batch,
// dataset/lidar_dataset.py:430-450).  Points sorted by their leaf-level voxel sit next to
// the points that share their node at every level, so one wave's gathers / scatter-adds hit the same rows and
// the fused kernel's run-length scatter issues one atomic per node run instead of one per point.
//
//   key[i]  = Z-order code of (quantize(coord[i], tree_level_world) - origin), same fp32 arithmetic as the query;
//             origin / per-axis bits come from the map's voxel bounding box, so the code has bx+by+bz
//             significant bits (23 for a 100 m x 16 m x 10 m street at 0.2 m) instead of 3*tree_level_world = 36
//             (only locality matters here: the fused kernel detects node runs by comparing corner ids)
//   perm    = argsort(key) by rocPRIM's radix sort over exactly those bits.
#include <cstring>
#include "shine_internal.hpp"

namespace shine {

struct SortBox {
  int ox, oy, oz;     // box corner, leaf voxel units
  int bx, by, bz;     // bits per axis
  int bmin;           // min(bx,by,bz): the cubic part that is bit-interleaved
};

// key = [x_hi | y_hi | z_hi | interleave3(x_lo, y_lo, z_lo)]: Z-order inside cubes of 2^bmin voxels, the cubes
// themselves in x-major order — bx+by+bz significant bits for an elongated (street-shaped) map.
template <typename K>
__global__ void k_sort_keys(const float* coord, long long n, float res, SortBox b, K* keys, unsigned long long* vals) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int vx = (int)quantize(coord[3 * i], res) - b.ox, vy = (int)quantize(coord[3 * i + 1], res) - b.oy,
      vz = (int)quantize(coord[3 * i + 2], res) - b.oz;
  // points outside the map's box (free-space samples beyond the last node) clamp to its faces
  const int lx = (1 << b.bx) - 1, ly = (1 << b.by) - 1, lz = (1 << b.bz) - 1;
  unsigned ux = (unsigned)(vx < 0 ? 0 : (vx > lx ? lx : vx)), uy = (unsigned)(vy < 0 ? 0 : (vy > ly ? ly : vy)),
           uz = (unsigned)(vz < 0 ? 0 : (vz > lz ? lz : vz));
  const unsigned m = (1u << b.bmin) - 1u;
  unsigned long long key = morton3(ux & m, uy & m, uz & m);
  int sh = 3 * b.bmin;
  key |= (unsigned long long)(uz >> b.bmin) << sh;
  sh += b.bz - b.bmin;
  key |= (unsigned long long)(uy >> b.bmin) << sh;
  sh += b.by - b.bmin;
  key |= (unsigned long long)(ux >> b.bmin) << sh;
  keys[i] = (K)key;
  vals[i] = (unsigned long long)i;
}

__global__ void k_sort_perm(const unsigned long long* vals, long long n, int* perm) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) perm[i] = (int)vals[i];
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// The radix sort is the library's one (u64 key, u64 value) instantiation (shine_prims.hip): rounds 1-2 carried four sort
// configurations here — 3.5 MB of code object for a function the hot path no longer calls (a batch is ordered by the
// counting sort of shine_plan.hip, ~2x faster than any radix sort at these sizes).
template <typename K>
static int sort_impl(const float* coord, long long n, float res, const SortBox& box, int32_t* perm_out,
                     void* workspace, size_t* workspace_bytes, hipStream_t st) {
  const unsigned end_bit = (unsigned)(box.bx + box.by + box.bz);
  const size_t cnt = (size_t)(n > 0 ? n : 1);
  size_t tmp_bytes = 0;
  SHINE_HIP_CHECK(prim_sort_pairs_u64(nullptr, tmp_bytes, nullptr, nullptr, nullptr, nullptr, cnt, 0u, end_bit, st));
  const size_t kb = align256(cnt * sizeof(K)), vb = align256(cnt * 8);
  const size_t need = 2 * kb + 2 * vb + align256(tmp_bytes);
  if (!workspace) {
    *workspace_bytes = need;
    return SHINE_OK;
  }
  if (*workspace_bytes < need) return set_error(SHINE_E_INVALID, "shine_morton_sort: workspace too small");
  if (n == 0) return SHINE_OK;
  char* w = (char*)workspace;
  K* k0 = (K*)w;
  K* k1 = (K*)(w + kb);
  unsigned long long* v0 = (unsigned long long*)(w + 2 * kb);
  unsigned long long* v1 = (unsigned long long*)(w + 2 * kb + vb);
  void* tmp = w + 2 * kb + 2 * vb;
  hipLaunchKernelGGL((k_sort_keys<K>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, coord, n, res, box, k0, v0);
  SHINE_HIP_CHECK(hipGetLastError());
  SHINE_HIP_CHECK(prim_sort_pairs_u64(tmp, tmp_bytes, k0, k1, v0, v1, (size_t)n, 0u, end_bit, st));
  hipLaunchKernelGGL(k_sort_perm, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, v1, n, (int*)perm_out);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

}  // namespace shine

using namespace shine;

// workspace layout: keys_in | keys_out | vals_in | vals_out | rocPRIM temp
extern "C" int shine_morton_sort(const shine_step_config* cfg, const float* coord, int64_t n, int32_t* perm_out,
                                 void* workspace, size_t* workspace_bytes, void* stream) {
  if (!cfg || !workspace_bytes || n < 0) return set_error(SHINE_E_INVALID, "shine_morton_sort: null argument");
  if (n > 0x7fffffffll) return set_error(SHINE_E_INVALID, "shine_morton_sort: n exceeds int32 permutation range");
  if (cfg->max_level < 1 || cfg->max_level > 15) return set_error(SHINE_E_INVALID, "shine_morton_sort: max_level");
  if (workspace && n > 0 && (!coord || !perm_out)) return set_error(SHINE_E_INVALID, "shine_morton_sort: null coord/perm");
  SortBox box = {cfg->sort_origin[0], cfg->sort_origin[1], cfg->sort_origin[2],
                 cfg->sort_bits[0],   cfg->sort_bits[1],   cfg->sort_bits[2], 0};
  const int ml = cfg->max_level;
  if (box.bx <= 0 || box.by <= 0 || box.bz <= 0 || box.bx > ml || box.by > ml || box.bz > ml) {
    box = SortBox{0, 0, 0, ml, ml, ml, 0};  // no box given: the whole cube
  }
  box.bmin = box.bx < box.by ? (box.bx < box.bz ? box.bx : box.bz) : (box.by < box.bz ? box.by : box.bz);
  const float res = (float)(1u << ml);
  hipStream_t st = (hipStream_t)stream;
  return sort_impl<unsigned long long>(coord, n, res, box, perm_out, workspace, workspace_bytes, st);
}
