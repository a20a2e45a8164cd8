// shine_sort.hip — Morton ordering of a point batch (new step right after LiDARDataset.get_batch,
// dataset/lidar_dataset.py:430-450).  Points sorted by their leaf-level node key sit
// next to the points that share their voxel, so one wave's gathers / scatter-adds hit the same rows.
//
//   keys[i] = Morton code of quantize(coord[i], tree_level_world)   (same arithmetic as the query itself)
//   perm    = argsort(keys)  via rocPRIM radix sort on the 3*level significant bits only
#include <hipcub/hipcub.hpp>

#include "shine_internal.hpp"

namespace shine {

__global__ void k_morton_keys(const float* coord, long long n, float res, unsigned long long* keys, int* vals) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x0 = coord[3 * i], x1 = coord[3 * i + 1], x2 = coord[3 * i + 2];
  keys[i] = morton3(quantize(x0, res), quantize(x1, res), quantize(x2, res));
  vals[i] = (int)i;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace shine

using namespace shine;

// workspace layout: keys_in | keys_out | vals_in | cub temp
extern "C" int shine_morton_sort(const shine_step_config* cfg, const float* coord, int64_t n, int32_t* perm_out,
                                 void* workspace, size_t* workspace_bytes, void* stream) {
  if (!cfg || !workspace_bytes || n < 0) return set_error(SHINE_E_INVALID, "shine_morton_sort: null argument");
  if (n > 0x7fffffffll) return set_error(SHINE_E_INVALID, "shine_morton_sort: n exceeds int32 permutation range");
  const int end_bit = 3 * cfg->max_level;
  size_t cub_bytes = 0;
  hipcub::DoubleBuffer<unsigned long long> dk(nullptr, nullptr);
  hipcub::DoubleBuffer<int> dv(nullptr, nullptr);
  SHINE_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, dk, dv, (int)(n > 0 ? n : 1), 0, end_bit,
                                                     (hipStream_t)stream));
  const size_t kb = align256((size_t)(n > 0 ? n : 1) * 8), vb = align256((size_t)(n > 0 ? n : 1) * 4);
  const size_t need = 2 * kb + vb + align256(cub_bytes);
  if (!workspace) {
    *workspace_bytes = need;
    return SHINE_OK;
  }
  if (*workspace_bytes < need) return set_error(SHINE_E_INVALID, "shine_morton_sort: workspace too small");
  if (n == 0) return SHINE_OK;
  if (!coord || !perm_out) return set_error(SHINE_E_INVALID, "shine_morton_sort: null coord/perm");
  char* w = (char*)workspace;
  unsigned long long* k0 = (unsigned long long*)w;
  unsigned long long* k1 = (unsigned long long*)(w + kb);
  int* v0 = (int*)(w + 2 * kb);
  void* tmp = w + 2 * kb + vb;
  hipStream_t st = (hipStream_t)stream;
  const float res = (float)(1u << cfg->max_level);
  hipLaunchKernelGGL(k_morton_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, coord, (long long)n, res, k0,
                     v0);
  SHINE_HIP_CHECK(hipGetLastError());
  hipcub::DoubleBuffer<unsigned long long> keys(k0, k1);
  hipcub::DoubleBuffer<int> vals(v0, (int*)perm_out);
  SHINE_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(tmp, cub_bytes, keys, vals, (int)n, 0, end_bit, st));
  if (vals.Current() != (int*)perm_out)
    SHINE_HIP_CHECK(hipMemcpyAsync(perm_out, vals.Current(), (size_t)n * 4, hipMemcpyDeviceToDevice, st));
  return SHINE_OK;
}
