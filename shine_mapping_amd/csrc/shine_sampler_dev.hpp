// shine_sampler_dev.hpp — device side of the sorted sampler shared by shine_sampler.hip (its own launches) and shine_finish.hip
// (the draw of the next iteration inside the optimiser's launch).  See shine_sampler.hip for the algorithm.
#pragma once
#include "shine_internal.hpp"

namespace shine {

// counter-based generator: splitmix64 finaliser of (seed, stream, counter)
// Exp(1) variate of draw k: -ln(u) with u on the 2^24 grid of (0,1], through the hardware fp32 log2 (v_log_f32, ~1 ulp)
// — a fp64 log() is a ~1000-cycle dependent chain and was the whole cost of a small draw (11 us for 4096 draws).  The
// running sums stay fp64; 24-bit variates truncate the exponential's tail at 16.6 (probability 6e-8).
__device__ __forceinline__ double exp1v(unsigned long long seed, unsigned long long stream, unsigned long long k) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (stream * 0x100000001B3ull + k + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const float u = ((float)(unsigned int)(z >> 40) + 1.0f) * (1.0f / 16777216.0f);
  return (double)(-0.693147180559945f * __builtin_amdgcn_logf(u));
}

constexpr int SB = 1024;  // draws per block (256 threads x 4)

__device__ __forceinline__ double block_sum_256(double v, double* s_red) {
  v = wave_sum_d(v);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  const double t = s_red[0] + s_red[1] + s_red[2] + s_red[3];
  __syncthreads();
  return t;
}

// Surface count of the batch (the eikonal term averages over the samples with weight > 0, shine_batch.py:183-185): the launch
// that writes the indices also counts its draws with weight > 0 (bit v of surf_bits: one bit per pool sample, so the 10^5-10^6
// look-ups of a draw hit a cache-resident array instead of gathering 4-byte words from the whole pool: 24 -> 7 us for pass 2
// at 2^20 draws from a 10^7-sample pool) into surf_parts, int64[SHINE_SURF_PARTS = 64] partial counts
// that the fused step adds up itself (one load per lane, cfg->n_surf_parts) — no launch of its own; in torch the same number
// costs six launches (index, compare, sum, ...: 45 us at 2^20 draws).  Two-launch form: pass 1 clears the 64 parts, block j of
// pass 2 adds its count to part j % 64 with one relaxed atomic (<= 16-fold contention per address for 2^20 draws; ONE shared
// counter would serialise ~1000 same-address atomics, and one part per block would make every wave of the step read ~1000
// words: 17 dependent round trips in its prologue, measured +13 us).  One-launch form (<= 16 blocks): block j stores part j.
constexpr int SURF_PARTS = 64;

__device__ __forceinline__ void block_count_256(int v, int* s_cnt, long long* surf_parts, int part, bool plain_store) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long c = (long long)(s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]);
    if (plain_store) surf_parts[part] = c;
    else __hip_atomic_fetch_add(surf_parts + part, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Small draws (the reference's bs = 4096 is 5 blocks): ONE launch.  Every block first recomputes all the block sums
// itself (at most FUSED_MAX_BLOCKS x 4 variates per thread, same arithmetic and summation order as pass 1, so the draw is
// bit-identical to the two-launch form), then does pass 2's work for its own block.  One graph node less per iteration —
// and as a device function (256 threads = virtual block `vb` of `nblocks`) it also rides in the optimiser's launch
// (shine_finish_iteration: the draw of the NEXT iteration), which leaves no launch of its own at all.
constexpr int FUSED_MAX_BLOCKS = 16;

struct SampleShared {
  double red[4];
  double wave_pre[4];
  double bs[FUSED_MAX_BLOCKS];
  int cnt[4];
};

__device__ __forceinline__ void sample_fused_block(SampleShared& sm, int vb, int nblocks, long long n, long long pool,
                                                   unsigned long long seed, unsigned long long stream,
                                                   unsigned long long* stream_dev, int* idx, const unsigned int* surf_bits,
                                                   long long* surf_parts) {
  if (stream_dev) stream = stream_dev[0];
  const long long n1 = n + 1;
  for (int b = 0; b < nblocks; ++b) {
    const long long k0 = (long long)b * SB + threadIdx.x * 4;
    double v = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + j < n1) v += exp1v(seed, stream, (unsigned long long)(k0 + j));
    const double t = block_sum_256(v, sm.red);
    if (threadIdx.x == 0) sm.bs[b] = t;
  }
  __syncthreads();
  double before = 0.0, total = 0.0;
  if ((int)threadIdx.x < nblocks) {
    const double v = sm.bs[threadIdx.x];
    total += v;
    if ((int)threadIdx.x < vb) before += v;
  }
  before = block_sum_256(before, sm.red);
  total = block_sum_256(total, sm.red);
  const long long k0 = (long long)vb * SB + threadIdx.x * 4;
  double e[4], run = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    e[j] = (k0 + j <= n) ? exp1v(seed, stream, (unsigned long long)(k0 + j)) : 0.0;
    run += e[j];
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double inc = run;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  if (lane == 63) sm.wave_pre[wv] = inc;
  __syncthreads();
  double wpre = 0.0;
  for (int w = 0; w < wv; ++w) wpre += sm.wave_pre[w];
  double sacc = before + wpre + (inc - run);
  int surf = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sacc += e[j];
    if (k0 + j < n) {
      long long v = (long long)((sacc / total) * (double)pool);
      v = v < 0 ? 0 : (v >= pool ? pool - 1 : v);
      idx[k0 + j] = (int)v;
      if (surf_parts) surf += (int)((surf_bits[v >> 5] >> (v & 31)) & 1u);
    }
  }
  if (surf_parts) {
    block_count_256(surf, sm.cnt, surf_parts, vb, true);
    if (vb == 0 && (int)threadIdx.x >= nblocks && threadIdx.x < SURF_PARTS) surf_parts[threadIdx.x] = 0;
  }
  if (stream_dev) {  // the last block to finish advances the stream id (<= 16 blocks: the counter costs < 1 us here)
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(&stream_dev[1], 1ull) == (unsigned long long)nblocks - 1ull) {
        stream_dev[1] = 0ull;
        stream_dev[2] = stream + 1ull;  // (keeps the shadow of the two-launch form in step)
        __threadfence();
        atomicAdd(&stream_dev[0], 1ull);
      }
    }
  }
}

}  // namespace shine
