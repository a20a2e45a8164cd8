// shine_exchange.hip — device side of the data-parallel "touched rows" gradient exchange (exchange only the
// feature rows a step touched instead of the dense tables).  The reference is single-GPU (utils/tools.py:26): no
// counterpart there.  Per step and rank:
//   shine_touched_index   byte flags (shine_mark_touched, OR-reduced over the ranks) -> ascending row-id lists per level
//                         (exclusive scan + compaction: the same order on every rank) and their counts
//   shine_touched_pack    the listed rows of every level's dense gradient, then the L trash rows -> one message
//   [RCCL all-reduce of the message, torch.distributed]
//   shine_touched_unpack  the reverse, and the flags are cleared for the next step
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "shine_internal.hpp"

namespace shine {

struct FlagToInt {
  __host__ __device__ int operator()(unsigned char f) const { return f ? 1 : 0; }
};

__global__ void k_touched_compact(const unsigned char* flags, const int* pos, long long n, int* idx_out,
                                  long long* count_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = flags[i] ? 1 : 0;
  if (f) idx_out[pos[i]] = (int)i;
  if (i == n - 1) *count_out = (long long)pos[i] + f;
}

struct ExchangeArgs {
  float* grad[SHINE_MAX_LEVELS];        // dense gradient tables [rows + 1][F]
  const int* idx[SHINE_MAX_LEVELS];     // ascending touched row ids
  unsigned char* flags[SHINE_MAX_LEVELS];
  long long off[SHINE_MAX_LEVELS + 1];  // first message row of each level; off[L] = total rows
  long long rows[SHINE_MAX_LEVELS];     // trash row index of each level
  int n_levels;
};

// one thread per (message row, feature): rows of level l occupy message rows [off[l], off[l+1]); the L trash rows follow
template <bool PACK>
__global__ void k_touched_move(ExchangeArgs a, float* msg) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long r = t >> 3;
  const int q = (int)(t & 7);
  const long long total = a.off[a.n_levels];
  if (r >= total + a.n_levels) return;
  float* g;
  if (r >= total) {  // trash rows: every miss of the step lands there
    const int l = (int)(r - total);
    g = a.grad[l] + a.rows[l] * F + q;
  } else {
    int l = 0;
#pragma unroll
    for (int k = 1; k < SHINE_MAX_LEVELS; ++k)
      if (k < a.n_levels && r >= a.off[k]) l = k;
    const int row = a.idx[l][r - a.off[l]];
    g = a.grad[l] + (long long)row * F + q;
    if (!PACK && q == 0 && a.flags[l]) a.flags[l][row] = 0;  // cleared for the next step
  }
  if (PACK) msg[t] = *g;
  else *g = msg[t];
}

static size_t xalign(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace shine

using namespace shine;

extern "C" int shine_touched_index(int32_t n_levels, const uint8_t* const* flags, const int64_t* rows,
                                   int32_t* const* idx_out, int64_t* counts_dev, void* workspace, size_t* workspace_bytes,
                                   void* stream) {
  if (n_levels < 1 || n_levels > SHINE_MAX_LEVELS || !rows || !workspace_bytes)
    return set_error(SHINE_E_INVALID, "shine_touched_index: bad argument");
  hipStream_t st = (hipStream_t)stream;
  long long max_rows = 0;
  for (int l = 0; l < n_levels; ++l) max_rows = rows[l] > max_rows ? rows[l] : max_rows;
  size_t scan_bytes = 0;
  auto it0 = rocprim::make_transform_iterator((const unsigned char*)nullptr, FlagToInt());
  SHINE_HIP_CHECK(rocprim::exclusive_scan(nullptr, scan_bytes, it0, (int*)nullptr, 0, (size_t)(max_rows > 0 ? max_rows : 1),
                                          rocprim::plus<int>(), st));
  const size_t need = xalign(scan_bytes) + xalign((size_t)max_rows * sizeof(int));
  if (!workspace) {
    *workspace_bytes = need;
    return SHINE_OK;
  }
  if (*workspace_bytes < need) return set_error(SHINE_E_INVALID, "shine_touched_index: workspace too small");
  if (!flags || !idx_out || !counts_dev) return set_error(SHINE_E_INVALID, "shine_touched_index: null argument");
  char* tmp = (char*)workspace;
  int* pos = (int*)(tmp + xalign(scan_bytes));
  for (int l = 0; l < n_levels; ++l) {
    const long long n = rows[l];
    if (n <= 0) {
      SHINE_HIP_CHECK(hipMemsetAsync(counts_dev + l, 0, sizeof(int64_t), st));
      continue;
    }
    auto it = rocprim::make_transform_iterator((const unsigned char*)flags[l], FlagToInt());
    size_t sb = scan_bytes;
    SHINE_HIP_CHECK(rocprim::exclusive_scan(tmp, sb, it, pos, 0, (size_t)n, rocprim::plus<int>(), st));
    hipLaunchKernelGGL(k_touched_compact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const unsigned char*)flags[l], (const int*)pos, n, (int*)idx_out[l], (long long*)(counts_dev + l));
    SHINE_HIP_CHECK(hipGetLastError());
  }
  return SHINE_OK;
}

static int touched_move(bool pack, int32_t n_levels, float* const* grads, const int32_t* const* idx,
                        const int64_t* counts, const int64_t* rows, uint8_t* const* flags, float* msg, void* stream) {
  if (n_levels < 1 || n_levels > SHINE_MAX_LEVELS || !grads || !idx || !counts || !rows || !msg)
    return set_error(SHINE_E_INVALID, "shine_touched_pack/unpack: bad argument");
  ExchangeArgs a = {};
  a.n_levels = n_levels;
  long long off = 0;
  for (int l = 0; l < n_levels; ++l) {
    a.grad[l] = grads[l];
    a.idx[l] = idx[l];
    a.flags[l] = flags ? flags[l] : nullptr;
    a.rows[l] = rows[l];
    a.off[l] = off;
    off += counts[l];
  }
  a.off[n_levels] = off;
  const long long threads = (off + n_levels) * F;
  const dim3 grid((unsigned)((threads + 255) / 256));
  if (pack) hipLaunchKernelGGL(k_touched_move<true>, grid, dim3(256), 0, (hipStream_t)stream, a, msg);
  else hipLaunchKernelGGL(k_touched_move<false>, grid, dim3(256), 0, (hipStream_t)stream, a, msg);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_touched_pack(int32_t n_levels, float* const* grads, const int32_t* const* idx, const int64_t* counts,
                                  const int64_t* rows, float* msg, void* stream) {
  return touched_move(true, n_levels, grads, idx, counts, rows, nullptr, msg, stream);
}

extern "C" int shine_touched_unpack(int32_t n_levels, float* const* grads, const int32_t* const* idx, const int64_t* counts,
                                    const int64_t* rows, uint8_t* const* flags, const float* msg, void* stream) {
  return touched_move(false, n_levels, grads, idx, counts, rows, flags, const_cast<float*>(msg), stream);
}
