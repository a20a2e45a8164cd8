// shine_sampler.hip — draw a training batch as SORTED i.i.d. indices into a node-ordered sample pool
// (LiDARDataset.get_batch, dataset/lidar_dataset.py:430-450, fused with the batch ordering).
//
// The reference draws `index = torch.randint(0, pool, (bs,))` and gathers.  The fused step wants its points in node
// order.  If the POOL is kept in node order (shine_plan_batch on the whole pool, once per frame), a batch is in node
// order as soon as its indices are sorted — and sorted i.i.d. uniform indices can be generated directly, without a
// sort, from the order statistics of the uniform distribution:
//       E_0..E_n ~ Exp(1) i.i.d. (exp1v below),  S_k = E_0 + .. + E_k,   U_(k) = S_k / S_n   (k < n)  are the sorted uniforms,
//       idx_k = floor(U_(k) * pool_size).
// As a multiset this is exactly `randint` (sampling with replacement); only the order differs, which no loss term
// depends on.  Two launches (block sums, then regenerate + scan + scale) — one for draws of up to 16 blocks — no atomics
// on the data path, no scratch array.
#include "shine_sampler_dev.hpp"

namespace shine {

// Two launches, no scratch array: the generator is counter-based, so pass 2 simply REGENERATES the variates of its
// block instead of reading them back.
//   pass 1  per block of SB draws: sum of its Exp(1) variates -> block_sum[b]   (also clears the gradient bucket)
//   pass 2  every block adds up the (few hundred) block sums in front of it, regenerates its variates, scans them in
//           LDS and writes idx_k = floor(S_k / S_total * pool).
// stream_dev != nullptr: the stream id is read from device memory (graph-replayable form, shine_sample_sorted_dev):
// uint64[4] = {A = stream id, block counter of the one-launch form, B = shadow of A, reserved}.  The two-launch form advances
// the id without atomics: pass 1 reads A and its block 0 stores B = A + 1 (nothing in pass 1 reads B); pass 2 reads B - 1
// and its block 0 stores A = B (nothing in pass 2 reads A) — the kernel boundaries order it.  (A "last block done" counter
// costs one fenced same-address atomic per block: 22-40 ns each, serialised — 41 us for the 1025 blocks of a 2^20 draw.)
__global__ __launch_bounds__(256) void k_sample_pass1(double* block_sum, long long n1, unsigned long long seed,
                                                      unsigned long long stream, const unsigned long long* stream_dev,
                                                      float4* zero_ptr, long long zero_n16, long long* surf_parts) {
  __shared__ double s_red[4];
  if (surf_parts && blockIdx.x == 0 && threadIdx.x < SURF_PARTS) surf_parts[threadIdx.x] = 0;  // pass 2 adds to them
  if (stream_dev) {
    stream = stream_dev[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) const_cast<unsigned long long*>(stream_dev)[2] = stream + 1ull;
  }
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  // ride-along clear of the gradient bucket (opt.zero_grad for the fused step), as in shine_plan_batch
  for (long long z = g; z < zero_n16; z += (long long)gridDim.x * 256) zero_ptr[z] = make_float4(0.f, 0.f, 0.f, 0.f);
  const long long k0 = (long long)blockIdx.x * SB + threadIdx.x * 4;
  double v = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (k0 + j < n1) v += exp1v(seed, stream, (unsigned long long)(k0 + j));
  const double t = block_sum_256(v, s_red);
  if (threadIdx.x == 0) block_sum[blockIdx.x] = t;
}

// [lo, lo + cnt): the part of the n sorted draws this launch writes (idx[k - lo]); block b of the launch is block
// blk0 + b of the whole draw.  A data-parallel rank draws only its contiguous slice of the GLOBAL batch this way: pass 1
// still covers all n + 1 spacings (compute only), pass 2 only the slice's blocks.
// FINISH = the second half of a draw whose pass 1 ran inside the previous fused step's reduction launch (cfg->next_draw): the
// same kernel, plus the ride-along clear of the gradient bucket that pass 1 does in the two-launch form.  (It trusts the block
// sums: the caller guarantees the rider ran for this draw — SortedPool.draw(pass1_done=True) checks it on the host.)
template <bool FINISH>
__global__ __launch_bounds__(256) void k_sample_pass2(const double* block_sum, int nblocks, long long n, long long pool,
                                                      unsigned long long seed, unsigned long long stream,
                                                      unsigned long long* stream_dev, int* idx, int blk0, long long lo,
                                                      long long cnt, const unsigned int* surf_bits, long long* surf_parts,
                                                      float4* zero_ptr, long long zero_n16) {
  __shared__ double s_red[4];
  __shared__ double s_wave_pre[4];
  __shared__ int s_cnt[4];
  if (stream_dev) {
    stream = stream_dev[2] - 1ull;
    if (blockIdx.x == 0 && threadIdx.x == 0) stream_dev[0] = stream + 1ull;
  }
  if (FINISH) {
    for (long long z = (long long)blockIdx.x * 256 + threadIdx.x; z < zero_n16; z += (long long)gridDim.x * 256)
      zero_ptr[z] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // prefix of the blocks in front of this one, and the grand total (nblocks is a few hundred)
  double before = 0.0, total = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    const double v = block_sum[b];
    total += v;
    if (b < blk0 + (int)blockIdx.x) before += v;
  }
  before = block_sum_256(before, s_red);
  total = block_sum_256(total, s_red);
  const long long k0 = (long long)(blk0 + blockIdx.x) * SB + threadIdx.x * 4;
  double e[4], run = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    e[j] = (k0 + j <= n) ? exp1v(seed, stream, (unsigned long long)(k0 + j)) : 0.0;
    run += e[j];
  }
  // exclusive scan of the per-thread sums across the block: within the wave by shuffles, across waves via LDS
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double inc = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) s_wave_pre[wv] = inc;
  __syncthreads();
  double wpre = 0.0;
  for (int w = 0; w < wv; ++w) wpre += s_wave_pre[w];
  double s = before + wpre + (inc - run);
  int surf = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    s += e[j];
    if (k0 + j < n && k0 + j >= lo && k0 + j < lo + cnt) {
      long long v = (long long)((s / total) * (double)pool);
      v = v < 0 ? 0 : (v >= pool ? pool - 1 : v);
      idx[k0 + j - lo] = (int)v;
      if (surf_parts) surf += (int)((surf_bits[v >> 5] >> (v & 31)) & 1u);
    }
  }
  if (surf_parts) block_count_256(surf, s_cnt, surf_parts, (blk0 + (int)blockIdx.x) & (SURF_PARTS - 1), false);
}

__global__ __launch_bounds__(256) void k_sample_fused(int nblocks, long long n, long long pool, unsigned long long seed,
                                                      unsigned long long stream, unsigned long long* stream_dev, int* idx,
                                                      float4* zero_ptr, long long zero_n16, const unsigned int* surf_bits,
                                                      long long* surf_parts) {
  __shared__ SampleShared sm;
  const long long gt = (long long)blockIdx.x * 256 + threadIdx.x;
  for (long long z = gt; z < zero_n16; z += (long long)gridDim.x * 256) zero_ptr[z] = make_float4(0.f, 0.f, 0.f, 0.f);
  sample_fused_block(sm, (int)blockIdx.x, nblocks, n, pool, seed, stream, stream_dev, idx, surf_bits, surf_parts);
}

static size_t align256s(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace shine

using namespace shine;

static int sample_sorted_impl(int64_t pool_size, int64_t n, uint64_t seed, uint64_t stream_id,
                              unsigned long long* stream_dev, int32_t* idx_out, void* zero_ptr, size_t zero_bytes,
                              const uint32_t* surf_bits, int64_t* surf_parts, void* workspace, size_t* workspace_bytes,
                              void* stream, int64_t slice_begin = 0, int64_t slice_n = -1) {
  if (slice_n < 0) slice_n = n - slice_begin;
  if (!workspace_bytes || n < 0 || pool_size < 1 || pool_size > 0x7fffffffll || slice_begin < 0 || slice_n < 0 ||
      slice_begin + slice_n > n)
    return set_error(SHINE_E_INVALID, "shine_sample_sorted: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long long n1 = (long long)n + 1;  // n draws + the closing spacing
  const long long nblocks = (n1 + SB - 1) / SB;
  const size_t need = align256s((size_t)nblocks * sizeof(double));
  if (!workspace) {
    *workspace_bytes = need;
    return SHINE_OK;
  }
  if (*workspace_bytes < need) return set_error(SHINE_E_INVALID, "shine_sample_sorted: workspace too small");
  if (zero_ptr && (((size_t)zero_ptr | zero_bytes) & 15))
    return set_error(SHINE_E_INVALID, "shine_sample_sorted: zero buffer must be 16-byte aligned and sized");
  // pass 2 over the blocks that hold the slice (an empty slice still runs one block: it advances the device stream id)
  const long long b0 = slice_n > 0 ? slice_begin / SB : 0;
  const long long b1 = slice_n > 0 ? (slice_begin + slice_n - 1) / SB : 0;
  if (surf_parts && !surf_bits) return set_error(SHINE_E_INVALID, "shine_sample_sorted: surf_parts needs surf_bits");
  if (n == 0) {
    if (zero_ptr && zero_bytes) SHINE_HIP_CHECK(hipMemsetAsync(zero_ptr, 0, zero_bytes, st));
    if (surf_parts) SHINE_HIP_CHECK(hipMemsetAsync(surf_parts, 0, (size_t)SURF_PARTS * 8, st));
    return SHINE_OK;
  }
  if (!idx_out) return set_error(SHINE_E_INVALID, "shine_sample_sorted: null output");
  if (nblocks <= FUSED_MAX_BLOCKS && slice_begin == 0 && slice_n == n) {
    hipLaunchKernelGGL(k_sample_fused, dim3((unsigned)nblocks), dim3(256), 0, st, (int)nblocks, (long long)n,
                       (long long)pool_size, (unsigned long long)seed, (unsigned long long)stream_id, stream_dev,
                       (int*)idx_out, (float4*)zero_ptr, zero_ptr ? (long long)(zero_bytes / 16) : 0ll, (const unsigned int*)surf_bits,
                       (long long*)surf_parts);
    SHINE_HIP_CHECK(hipGetLastError());
    return SHINE_OK;
  }
  double* bs = (double*)workspace;
  hipLaunchKernelGGL(k_sample_pass1, dim3((unsigned)nblocks), dim3(256), 0, st, bs, n1, (unsigned long long)seed,
                     (unsigned long long)stream_id, (const unsigned long long*)stream_dev, (float4*)zero_ptr,
                     zero_ptr ? (long long)(zero_bytes / 16) : 0ll, (long long*)surf_parts);
  SHINE_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_sample_pass2<false>, dim3((unsigned)(b1 - b0 + 1)), dim3(256), 0, st, bs, (int)nblocks, (long long)n,
                     (long long)pool_size, (unsigned long long)seed, (unsigned long long)stream_id, stream_dev,
                     (int*)idx_out, (int)b0, (long long)slice_begin, (long long)slice_n, (const unsigned int*)surf_bits, (long long*)surf_parts,
                     (float4*)nullptr, 0ll);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

// Data-parallel draw: rank r wants draws [slice_begin, slice_begin + slice_n) of the ONE global sorted batch of n draws
// every rank agrees on (same seed and stream id everywhere).  idx_out [slice_n].  stream_state: device uint64[2] as in
// shine_sample_sorted_dev, or NULL to use stream_id.
extern "C" int shine_sample_sorted_slice(int64_t pool_size, int64_t n, int64_t slice_begin, int64_t slice_n, uint64_t seed,
                                         uint64_t stream_id, uint64_t* stream_state, int32_t* idx_out, void* zero_ptr,
                                         size_t zero_bytes, const uint32_t* surf_bits, int64_t* surf_parts, void* workspace,
                                         size_t* workspace_bytes, void* stream) {
  return sample_sorted_impl(pool_size, n, seed, stream_id, (unsigned long long*)stream_state, idx_out, zero_ptr,
                            zero_bytes, surf_bits, surf_parts, workspace, workspace_bytes, stream, slice_begin, slice_n);
}

extern "C" int shine_sample_sorted(int64_t pool_size, int64_t n, uint64_t seed, uint64_t stream_id, int32_t* idx_out,
                                   void* zero_ptr, size_t zero_bytes, const uint32_t* surf_bits, int64_t* surf_parts,
                                   void* workspace, size_t* workspace_bytes, void* stream) {
  return sample_sorted_impl(pool_size, n, seed, stream_id, nullptr, idx_out, zero_ptr, zero_bytes, surf_bits, surf_parts,
                            workspace, workspace_bytes, stream);
}

extern "C" int shine_sample_sorted_dev(int64_t pool_size, int64_t n, uint64_t seed, uint64_t* stream_state,
                                       int32_t* idx_out, void* zero_ptr, size_t zero_bytes, const unsigned int* surf_bits,
                                       int64_t* surf_parts, void* workspace, size_t* workspace_bytes, void* stream) {
  if (workspace && !stream_state) return set_error(SHINE_E_INVALID, "shine_sample_sorted_dev: null stream_state");
  return sample_sorted_impl(pool_size, n, seed, 0, (unsigned long long*)stream_state, idx_out, zero_ptr, zero_bytes,
                            surf_bits, surf_parts, workspace, workspace_bytes, stream);
}

// The second half of a graph-replayable draw whose pass 1 rode on the previous shine_train_step (cfg->next_draw): ONE launch
// (index writing + surface count + the ride-along clear).  ONLY valid directly after such a step for the same draw size.
extern "C" int shine_sample_sorted_finish(int64_t pool_size, int64_t n, int64_t slice_begin, int64_t slice_n, uint64_t seed,
                                          uint64_t* stream_state, int32_t* idx_out, void* zero_ptr, size_t zero_bytes,
                                          const uint32_t* surf_bits, int64_t* surf_parts, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  if (slice_n < 0) slice_n = n - slice_begin;
  const long long nblocks = ((long long)n + 1 + SB - 1) / SB;
  if (n < 1 || pool_size < 1 || pool_size > 0x7fffffffll || slice_begin < 0 || slice_n < 0 || slice_begin + slice_n > n ||
      !stream_state || !idx_out || !workspace || workspace_bytes < (size_t)nblocks * sizeof(double) ||
      (surf_parts && !surf_bits))
    return set_error(SHINE_E_INVALID, "shine_sample_sorted_finish: bad argument (n >= 1, device stream state, the draw's workspace)");
  if (zero_ptr && (((size_t)zero_ptr | zero_bytes) & 15))
    return set_error(SHINE_E_INVALID, "shine_sample_sorted_finish: zero buffer must be 16-byte aligned and sized");
  const long long b0 = slice_n > 0 ? slice_begin / SB : 0;
  const long long b1 = slice_n > 0 ? (slice_begin + slice_n - 1) / SB : 0;
  hipLaunchKernelGGL(k_sample_pass2<true>, dim3((unsigned)(b1 - b0 + 1)), dim3(256), 0, (hipStream_t)stream,
                     (const double*)workspace, (int)nblocks, (long long)n, (long long)pool_size, (unsigned long long)seed, 0ull,
                     (unsigned long long*)stream_state, (int*)idx_out, (int)b0, (long long)slice_begin, (long long)slice_n,
                     (const unsigned int*)surf_bits, (long long*)surf_parts, (float4*)zero_ptr,
                     zero_ptr ? (long long)(zero_bytes / 16) : 0ll);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

// shine_draw_rider (include/shine_hip.h): set the chain of riding draws up — draw 0 with stream id `first_stream_id` into idx_out /
// surf_parts[0], pass 1 of draw 1 into block_sum[1], state[0] = first_stream_id, surf_parts[1] = 0.  The first step has parity 0.
namespace shine {
__global__ void k_set_u64(unsigned long long* p, unsigned long long v) { *p = v; }
}  // namespace shine

extern "C" int shine_draw_rider_prime(const shine_draw_rider* r, uint64_t first_stream_id, void* stream) {
  if (!r) return set_error(SHINE_E_INVALID, "shine_draw_rider_prime: null argument");
  const long long n1 = (long long)r->n + 1, nblocks = (n1 + SB - 1) / SB;
  if (r->n < 1 || r->pool_size < 1 || r->pool_size > 0x7fffffffll || !r->state || !r->idx_out || !r->block_sum[0] ||
      !r->block_sum[1] || (r->surf_bits && (!r->surf_parts[0] || !r->surf_parts[1])))
    return set_error(SHINE_E_INVALID, "shine_draw_rider_prime: bad argument");
  hipStream_t st = (hipStream_t)stream;
  long long* parts0 = r->surf_bits ? reinterpret_cast<long long*>(r->surf_parts[0]) : nullptr;
  // draw 0: the stand-alone two-launch form (pass 1 clears surf_parts[0], pass 2 adds to it), block sums through block_sum[0]
  hipLaunchKernelGGL(k_sample_pass1, dim3((unsigned)nblocks), dim3(256), 0, st, r->block_sum[0], n1, (unsigned long long)r->seed,
                     (unsigned long long)first_stream_id, (const unsigned long long*)nullptr, (float4*)nullptr, 0ll, parts0);
  hipLaunchKernelGGL(k_sample_pass2<false>, dim3((unsigned)(((long long)r->n - 1) / SB + 1)), dim3(256), 0, st,
                     (const double*)r->block_sum[0], (int)nblocks, (long long)r->n, (long long)r->pool_size,
                     (unsigned long long)r->seed, (unsigned long long)first_stream_id, (unsigned long long*)nullptr, (int*)r->idx_out, 0, 0ll,
                     (long long)r->n, (const unsigned int*)r->surf_bits, parts0, (float4*)nullptr, 0ll);
  // pass 1 of draw 1
  hipLaunchKernelGGL(k_sample_pass1, dim3((unsigned)nblocks), dim3(256), 0, st, r->block_sum[1], n1, (unsigned long long)r->seed,
                     (unsigned long long)first_stream_id + 1ull, (const unsigned long long*)nullptr, (float4*)nullptr, 0ll,
                     (long long*)nullptr);
  hipLaunchKernelGGL(k_set_u64, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned long long*>(r->state),
                     (unsigned long long)first_stream_id);
  if (r->surf_bits) SHINE_HIP_CHECK(hipMemsetAsync(r->surf_parts[1], 0, (size_t)SURF_PARTS * 8, st));
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}
