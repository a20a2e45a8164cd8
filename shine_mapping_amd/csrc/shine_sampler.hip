// shine_sampler.hip — draw a training batch as SORTED i.i.d. indices into a node-ordered sample pool
// (LiDARDataset.get_batch, dataset/lidar_dataset.py:430-450, fused with the batch ordering).
//
// The reference draws `index = torch.randint(0, pool, (bs,))` and gathers.  The fused step wants its points in node
// order.  If the POOL is kept in node order (shine_plan_batch on the whole pool, once per frame), a batch is in node
// order as soon as its indices are sorted — and sorted i.i.d. uniform indices can be generated directly, without a
// sort, from the order statistics of the uniform distribution:
//       E_0..E_n ~ Exp(1) i.i.d.,  S_k = E_0 + .. + E_k,   U_(k) = S_k / S_n   (k < n)  are the sorted uniforms,
//       idx_k = floor(U_(k) * pool_size).
// As a multiset this is exactly `randint` (sampling with replacement); only the order differs, which no loss term
// depends on.  Three launches: Exp(1) variates from a counter-based generator, an inclusive scan (rocPRIM), the
// scaling.  No atomics, ~n * 16 B of traffic.
#include <cstring>
#include <rocprim/device/device_scan.hpp>

#include "shine_internal.hpp"

namespace shine {

// counter-based generator: splitmix64 finaliser of (seed, stream, counter) -> uniform in (0,1]
__device__ __forceinline__ double u01(unsigned long long seed, unsigned long long stream, unsigned long long k) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (stream * 0x100000001B3ull + k + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return ((double)(z >> 11) + 1.0) * (1.0 / 9007199254740992.0);  // (0,1]
}

__global__ __launch_bounds__(256) void k_exp_variates(double* e, long long n1, unsigned long long seed,
                                                      unsigned long long stream, float4* zero_ptr, long long zero_n16) {
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  // ride-along clear of the gradient bucket (opt.zero_grad for the fused step), as in shine_plan_batch
  for (long long z = k; z < zero_n16; z += (long long)gridDim.x * 256) zero_ptr[z] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (k < n1) e[k] = -log(u01(seed, stream, (unsigned long long)k));
}

__global__ __launch_bounds__(256) void k_scale_indices(const double* s, long long n, long long pool, int* idx) {
  const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  const double u = s[k] / s[n];  // s[n] = total of the n+1 variates
  long long v = (long long)(u * (double)pool);
  idx[k] = (int)(v < 0 ? 0 : (v >= pool ? pool - 1 : v));
}

static size_t align256s(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace shine

using namespace shine;

extern "C" int shine_sample_sorted(int64_t pool_size, int64_t n, uint64_t seed, uint64_t stream_id, int32_t* idx_out,
                                   void* zero_ptr, size_t zero_bytes, void* workspace, size_t* workspace_bytes,
                                   void* stream) {
  if (!workspace_bytes || n < 0 || pool_size < 1 || pool_size > 0x7fffffffll)
    return set_error(SHINE_E_INVALID, "shine_sample_sorted: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t n1 = (size_t)n + 1;
  size_t scan_bytes = 0;
  SHINE_HIP_CHECK(rocprim::inclusive_scan(nullptr, scan_bytes, (double*)nullptr, (double*)nullptr, n1,
                                          rocprim::plus<double>(), st));
  const size_t o_e = 0, o_scan = align256s(n1 * 8), need = o_scan + align256s(scan_bytes);
  if (!workspace) {
    *workspace_bytes = need;
    return SHINE_OK;
  }
  if (*workspace_bytes < need) return set_error(SHINE_E_INVALID, "shine_sample_sorted: workspace too small");
  if (zero_ptr && (((size_t)zero_ptr | zero_bytes) & 15))
    return set_error(SHINE_E_INVALID, "shine_sample_sorted: zero buffer must be 16-byte aligned and sized");
  if (n == 0) {
    if (zero_ptr && zero_bytes) SHINE_HIP_CHECK(hipMemsetAsync(zero_ptr, 0, zero_bytes, st));
    return SHINE_OK;
  }
  if (!idx_out) return set_error(SHINE_E_INVALID, "shine_sample_sorted: null output");
  double* e = (double*)((char*)workspace + o_e);
  const dim3 grid((unsigned)((n1 + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_exp_variates, grid, block, 0, st, e, (long long)n1, (unsigned long long)seed,
                     (unsigned long long)stream_id, (float4*)zero_ptr, zero_ptr ? (long long)(zero_bytes / 16) : 0ll);
  SHINE_HIP_CHECK(hipGetLastError());
  SHINE_HIP_CHECK(rocprim::inclusive_scan((char*)workspace + o_scan, scan_bytes, e, e, n1, rocprim::plus<double>(), st));
  hipLaunchKernelGGL(k_scale_indices, grid, block, 0, st, e, (long long)n, (long long)pool_size, (int*)idx_out);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}
