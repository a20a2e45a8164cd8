// shine_prims.hip — the ONE translation unit that instantiates rocPRIM (device scan, radix sort).  Every caller (octree growth,
// batch plan, Morton sort, the data-parallel row exchanges) goes through these four host wrappers: rocPRIM's kernels are
// templates, so each translation unit that names them gets its own copy of the code objects — shine_grow / shine_sort /
// shine_plan / shine_exchange together carried 9 MB of duplicated scans and sorts (the .so < 6 MB).
// Same calling convention as rocPRIM: tmp == nullptr returns the temporary-storage size in `bytes`.
#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "shine_internal.hpp"

namespace shine {

struct FlagToInt {
  __host__ __device__ int operator()(unsigned char f) const { return f ? 1 : 0; }
};

// out[i] = sum of in[0..i) (int, in-place allowed)
hipError_t prim_scan_int(void* tmp, size_t& bytes, const int* in, int* out, size_t n, hipStream_t st) {
  return rocprim::exclusive_scan(tmp, bytes, in, out, 0, n, rocprim::plus<int>(), st);
}

// out[i] = number of non-zero bytes in flags[0..i)
hipError_t prim_scan_flags(void* tmp, size_t& bytes, const unsigned char* flags, int* out, size_t n, hipStream_t st) {
  auto it = rocprim::make_transform_iterator(flags, FlagToInt());
  return rocprim::exclusive_scan(tmp, bytes, it, out, 0, n, rocprim::plus<int>(), st);
}

// keys only: the (u64, u64) pair sort with two scratch value arrays in front of rocPRIM's temporary storage — a second
// radix-sort instantiation is ~2.5 MB of code object, the extra 16 B per key of traffic a few microseconds per frame
// (the callers sort <= a few 10^5 node / corner keys when the octree grows)
hipError_t prim_sort_keys_u64(void* tmp, size_t& bytes, const unsigned long long* k0, unsigned long long* k1, size_t n,
                              unsigned begin_bit, unsigned end_bit, hipStream_t st) {
  const size_t vb = (n * sizeof(unsigned long long) + 255) & ~(size_t)255;
  if (!tmp) {
    size_t inner = 0;
    const hipError_t e = rocprim::radix_sort_pairs(nullptr, inner, k0, k1, (const unsigned long long*)nullptr,
                                                   (unsigned long long*)nullptr, n, begin_bit, end_bit, st);
    bytes = inner + 2 * vb;
    return e;
  }
  if (bytes < 2 * vb) return hipErrorInvalidValue;
  unsigned long long* v0 = reinterpret_cast<unsigned long long*>(tmp);
  unsigned long long* v1 = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(tmp) + vb);
  size_t inner = bytes - 2 * vb;
  return rocprim::radix_sort_pairs(reinterpret_cast<char*>(tmp) + 2 * vb, inner, k0, k1, (const unsigned long long*)v0, v1, n,
                                   begin_bit, end_bit, st);
}

hipError_t prim_sort_pairs_u64(void* tmp, size_t& bytes, const unsigned long long* k0, unsigned long long* k1,
                               const unsigned long long* v0, unsigned long long* v1, size_t n, unsigned begin_bit,
                               unsigned end_bit, hipStream_t st) {
  return rocprim::radix_sort_pairs(tmp, bytes, k0, k1, v0, v1, n, begin_bit, end_bit, st);
}

// Segments of <= 4096 int32 values sorted ascending in place, one workgroup per segment (rocPRIM's block radix sort over the low
// `end_bit` bits: three or four ranking passes against the 78 barrier stages of a bitonic network): segment s = idx[s * seg ..
// min((s + 1) * seg, total)).  The importance sweep's chunks (shine_sweep.hip) — every shipped yaml has bs = 4096.
__global__ __launch_bounds__(1024) void k_sort_segments_4k(int* __restrict__ idx, long long seg, long long total, unsigned end_bit) {
  using sort_t = rocprim::block_radix_sort<unsigned int, 1024, 4>;
  __shared__ typename sort_t::storage_type storage;
  const long long first = (long long)blockIdx.x * seg;
  const long long cnt = total - first < seg ? total - first : seg;
  unsigned int keys[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = (long long)threadIdx.x * 4 + k;
    keys[k] = i < cnt ? (unsigned int)idx[first + i] : 0xffffffffu;  // (padding sorts behind every position)
  }
  sort_t().sort(keys, storage, 0u, end_bit);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = (long long)threadIdx.x * 4 + k;
    if (i < cnt) idx[first + i] = (int)keys[k];
  }
}

hipError_t prim_sort_segments_4k(int* idx, long long seg, long long total, int n_seg, unsigned end_bit, hipStream_t st) {
  if (seg > 4096 || n_seg < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_sort_segments_4k, dim3((unsigned)n_seg), dim3(1024), 0, st, idx, seg, total, end_bit);
  return hipGetLastError();
}

// rocPRIM picks its kernels by problem size (single-block sorts, merge sort, onesweep radix passes: a dozen code objects), and
// HIP loads a kernel's code on its first launch — tens of milliseconds the first time a frame's octree growth crosses into a
// size class (measured: one 40-68 ms frame in an incremental run whose other frames take 3.5 ms).  One pass over the size
// classes, once per process, in front of the first growth.
hipError_t prim_warmup(hipStream_t st) {
  static bool done = false;
  if (done) return hipSuccess;
  const size_t sizes[] = {64, 1024, 3000, 4096, 10000, 40000, 200000, 1500000};
  const size_t nmax = sizes[sizeof(sizes) / sizeof(sizes[0]) - 1];
  size_t sort_bytes = 0, scan_bytes = 0;
  hipError_t e = prim_sort_keys_u64(nullptr, sort_bytes, nullptr, nullptr, nmax, 0u, 64u, st);
  if (e != hipSuccess) return e;
  e = prim_scan_int(nullptr, scan_bytes, nullptr, nullptr, nmax, st);
  if (e != hipSuccess) return e;
  const size_t tb = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  char* buf = nullptr;
  e = hipMalloc(reinterpret_cast<void**>(&buf), 2 * nmax * 8 + tb);
  if (e != hipSuccess) return e;
  unsigned long long* k0 = reinterpret_cast<unsigned long long*>(buf);
  unsigned long long* k1 = k0 + nmax;
  e = hipMemsetAsync(buf, 0x5a, 2 * nmax * 8, st);
  for (size_t n : sizes) {
    for (unsigned end_bit : {36u, 45u, 64u}) {
      size_t sb = 0;
      if (e == hipSuccess) e = prim_sort_keys_u64(nullptr, sb, nullptr, nullptr, n, 0u, end_bit, st);
      if (e == hipSuccess) e = prim_sort_keys_u64(buf + 2 * nmax * 8, sb, k0, k1, n, 0u, end_bit, st);
    }
    size_t cb = 0;
    if (e == hipSuccess) e = prim_scan_int(nullptr, cb, nullptr, nullptr, n, st);
    if (e == hipSuccess) e = prim_scan_int(buf + 2 * nmax * 8, cb, reinterpret_cast<const int*>(k0), reinterpret_cast<int*>(k1), n, st);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(buf);
  done = e == hipSuccess;
  return e;
}

}  // namespace shine
