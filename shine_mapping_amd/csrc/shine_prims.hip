// shine_prims.hip — the ONE translation unit that instantiates rocPRIM (device scan, radix sort).  Every caller (octree growth,
// batch plan, Morton sort, the data-parallel row exchanges) goes through these four host wrappers: rocPRIM's kernels are
// templates, so each translation unit that names them gets its own copy of the code objects — shine_grow / shine_sort /
// shine_plan / shine_exchange together carried 9 MB of duplicated scans and sorts (the .so < 6 MB).
// Same calling convention as rocPRIM: tmp == nullptr returns the temporary-storage size in `bytes`.
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

}  // namespace shine
