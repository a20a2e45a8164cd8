// shine_plan.hip — order a batch by OCTREE NODE with a counting sort and hand the fused step each point's
// hash slots (replaces a general radix sort of Morton keys; new step after LiDARDataset.get_batch,
// dataset/lidar_dataset.py:430-450).
//
// Every node of every featured level has a rank in one global Z-order (a parent's own bucket sits right after
// its children's; shine_tables_rank_nodes sorts them on the device whenever the tree grows, or the caller uploads
// them with shine_tables_set_ranks).  A point's bucket is the rank of the DEEPEST node that contains it (or one of the 64
// trailing miss buckets if it misses at every level).  Then
//     k_plan_count    probe all L levels once (light kernel, 8 waves/SIMD hide the dependent loads), remember the
//                     slots, count the buckets with RETURNING atomics (the old value is the point's rank in its
//                     bucket); optionally clears the gradient bucket in the same pass;
//     exclusive scan  of the histogram (rocPRIM, single pass);
//     k_plan_scatter  position = offset[bucket] + rank ; perm[position] = i ; slots_sorted[position] = slots(i)
//                     (no atomics).
// Points of one node end up adjacent (what the fused kernel's run-length scatter needs), nodes follow the Z-order
// (L2 locality), and the heavy MFMA kernel no longer hashes or probes: it reads its slots coalesced.
// ~3 small launches instead of a histogram + 3..5 radix passes; order inside a bucket is arbitrary (atomics).
// Batches of <= 16384 points (the reference's batch size is 4096) take ONE launch and are not reordered: k_plan_unsorted.
#include <cstring>

#include "shine_internal.hpp"

namespace shine {

struct PlanLevel {
  const unsigned long long* keys;
  const int* ranks;
  unsigned int shift, mask;
};
struct PlanArgs {
  PlanLevel lv[SHINE_MAX_LEVELS];
  const float* coord;
  long long n;
  int n_levels;
  float res_leaf;
  int first_miss_bucket;  // the last MISS_BUCKETS buckets collect the points that miss at every level
  int* count;        // [n_buckets]
  int* bucket;       // [n]
  int* local;        // [n] rank of the point inside its bucket (returned by the counting atomic)
  float4* zero_ptr;  // optional buffer to clear in the same pass (the flat gradient bucket), 16-B units
  long long zero_n16;
  int ablate;  // debug (kernel_variant >> 8): 1 no counting atomics, 2 no probes (all miss), 4 no slot stores, 8 sort small batches too
  int* slots_tmp;    // [n][L]
  int* perm;         // [n]
  int* slots_sorted; // [n][L]
};

// Points that miss everywhere touch no table row, so their order is irrelevant — but one shared bucket would be a
// 10^4-way atomic hot spot.  They are spread over MISS_BUCKETS buckets (by wave id) and each wave adds its whole
// group with ONE atomic (ballot + popcount); node buckets hold a handful of points each and use plain atomics.
constexpr int MISS_BUCKETS = 64;

// probe all levels for point i: writes its hash slots (slots_tmp) and returns the rank of the deepest node that holds it (-1: none)
__device__ __forceinline__ int plan_probe(const PlanArgs& a, long long i) {
  int b = -1;
  const float x0 = a.coord[3 * i], x1 = a.coord[3 * i + 1], x2 = a.coord[3 * i + 2];
  const unsigned long long kleaf = morton3(quantize(x0, a.res_leaf), quantize(x1, a.res_leaf), quantize(x2, a.res_leaf));
  const int L = a.n_levels;
  // all levels' first-slot key loads and (speculative) rank loads are issued together: one memory round trip for the
  // common no-collision case instead of L serialized probe loops
  unsigned long long key[SHINE_MAX_LEVELS], k0[SHINE_MAX_LEVELS];
  unsigned int slot0[SHINE_MAX_LEVELS];
  int r0[SHINE_MAX_LEVELS];
#pragma unroll
  for (int s = 0; s < SHINE_MAX_LEVELS; ++s) {  // no guard: unused levels alias level 0's table (host side), loads are harmless
    const PlanLevel& Lv = a.lv[s];
    key[s] = s < L ? kleaf >> (3 * (L - 1 - s)) : 0ull;
    slot0[s] = hash_slot(key[s], Lv.shift);
    k0[s] = Lv.keys[slot0[s]];
    r0[s] = Lv.ranks[slot0[s]];
  }
#pragma unroll
  for (int s = 0; s < SHINE_MAX_LEVELS; ++s) {  // coarse -> fine: the deepest hit wins
    if (s < L) {
      const PlanLevel& Lv = a.lv[s];
      int found = -1, rk = -1;
      if (k0[s] == key[s]) {
        found = (int)slot0[s];
        rk = r0[s];
      } else if (k0[s] != EMPTY_KEY) {  // first-slot collision: walk the probe sequence (rare at load <= 0.5)
        unsigned int slot = (slot0[s] + 1) & Lv.mask;
        for (unsigned int n = 0; n < Lv.mask; ++n) {
          const unsigned long long k = Lv.keys[slot];
          if (k == key[s]) {
            found = (int)slot;
            rk = Lv.ranks[slot];
            break;
          }
          if (k == EMPTY_KEY) break;
          slot = (slot + 1) & Lv.mask;
        }
      }
      if (!(a.ablate & 4)) a.slots_tmp[i * L + s] = found;
      if (found >= 0) b = rk;
    }
  }
  return b;
}

__global__ __launch_bounds__(256) void k_plan_count(PlanArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  // ride-along memset of the gradient bucket (what opt.zero_grad amounts to): streaming stores overlap with this
  // kernel's dependent probe loads instead of costing a launch of their own
  for (long long z = i; z < a.zero_n16; z += (long long)gridDim.x * 256) a.zero_ptr[z] = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool live = i < a.n;
  int b = -1;
  if (live && !(a.ablate & 2)) b = plan_probe(a, i);
  const bool miss = live && b < 0;
  const unsigned long long mm = __ballot(miss);
  int loc = 0;
  if (mm) {  // the wave's miss group takes its ranks with one atomic
    const int leader = __ffsll((long long)mm) - 1;
    const int mb = a.first_miss_bucket + (int)((i >> 6) & (MISS_BUCKETS - 1));  // same for the whole wave
    int base = 0;
    if (lane == leader) base = atomicAdd(&a.count[mb], __popcll(mm));
    base = __shfl(base, leader, 64);
    if (miss) {
      b = mb;
      loc = base + __popcll(mm & ((1ull << lane) - 1ull));
    }
  }
  if (live && !miss && !(a.ablate & 1)) loc = atomicAdd(&a.count[b], 1);
  if (live) {
    a.bucket[i] = b;
    a.local[i] = loc;
  }
}

__global__ __launch_bounds__(256) void k_plan_scatter(PlanArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  const int L = a.n_levels;
  const int pos = a.count[a.bucket[i]] + a.local[i];  // count[] holds the exclusive offsets by now: no atomics
  a.perm[pos] = (int)i;
  for (int s = 0; s < L; ++s) a.slots_sorted[(long long)pos * L + s] = a.slots_tmp[i * L + s];
}

// The reference's batch size (4096 points: every shipped yaml; anything up to 16384) is planned WITHOUT the sort: at that size the
// fused step is a few tiles per wave and bound by latency, not by its atomics or the L2 — what the node order buys a 2^18-point batch (merged node runs,
// locality) is below the noise there, while the histogram over the tree's 10^5..10^7 node ranks costs four launches and ~25 us.
// One launch: probe the slots, perm = identity (deterministic), clear the gradient bucket.  profiles/r05_ab_experiments.txt block 9
// (a one-workgroup bitonic sort in LDS was measured first: 53-68 us on the device, VALU- and TA-bound on its single CU).
constexpr int PLAN_UNSORTED_MAX = 16384;  // (measured: the step does not see the order up to here; at 65536 it is 1.6x slower unsorted)

__global__ __launch_bounds__(256) void k_plan_unsorted(PlanArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (long long z = i; z < a.zero_n16; z += (long long)gridDim.x * 256) a.zero_ptr[z] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i >= a.n) return;
  plan_probe(a, i);  // (slots_tmp IS slots_sorted here)
  a.perm[i] = (int)i;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace shine

using namespace shine;

// workspace layout: count[n_buckets] | bucket[n] | local[n] | slots_tmp[n*L] | scan temp
extern "C" int shine_plan_batch(const shine_tables* t, const shine_step_config* cfg, const float* coord, int64_t n,
                                int32_t* perm_out, int32_t* slots_out, void* zero_ptr, size_t zero_bytes,
                                void* workspace, size_t* workspace_bytes, void* stream) {
  if (!t || !cfg || !workspace_bytes || n < 0) return set_error(SHINE_E_INVALID, "shine_plan_batch: null argument");
  if (n > 0x7fffffffll) return set_error(SHINE_E_INVALID, "shine_plan_batch: n exceeds int32 range");
  const int L = cfg->n_levels;
  if (L < 1 || L > SHINE_MAX_LEVELS || L != t->n_levels) return set_error(SHINE_E_INVALID, "shine_plan_batch: n_levels mismatch");
  if (t->n_buckets < MISS_BUCKETS + 1) return set_error(SHINE_E_STATE, "shine_plan_batch: node ranks not set (shine_tables_set_ranks)");
  hipStream_t st = (hipStream_t)stream;
  const size_t nb = (size_t)t->n_buckets, cnt = (size_t)(n > 0 ? n : 1);
  size_t scan_bytes = 0;
  SHINE_HIP_CHECK(prim_scan_int(nullptr, scan_bytes, nullptr, nullptr, nb, st));
  const size_t o_count = 0, o_bucket = o_count + align256(nb * 4), o_local = o_bucket + align256(cnt * 4),
               o_slots = o_local + align256(cnt * 4), o_scan = o_slots + align256(cnt * (size_t)L * 4),
               need = o_scan + align256(scan_bytes);
  if (zero_ptr && (((size_t)zero_ptr | zero_bytes) & 15))
    return set_error(SHINE_E_INVALID, "shine_plan_batch: zero buffer must be 16-byte aligned and sized");
  if (!workspace) {
    *workspace_bytes = need;
    return SHINE_OK;
  }
  if (*workspace_bytes < need) return set_error(SHINE_E_INVALID, "shine_plan_batch: workspace too small");
  if (n == 0) {
    if (zero_ptr && zero_bytes) SHINE_HIP_CHECK(hipMemsetAsync(zero_ptr, 0, zero_bytes, st));
    return SHINE_OK;
  }
  if (!coord || !perm_out || !slots_out) return set_error(SHINE_E_INVALID, "shine_plan_batch: null coord/perm/slots");
  char* w = (char*)workspace;
  PlanArgs a = {};
  for (int s = 0; s < L; ++s) {
    const TableLevel& T = t->lv[s];
    if (!T.keys || !T.ranks) return set_error(SHINE_E_STATE, "shine_plan_batch: a featured level has no table yet");
    a.lv[s].keys = T.keys;
    a.lv[s].ranks = T.ranks;
    a.lv[s].shift = T.shift;
    a.lv[s].mask = T.mask;
  }
  for (int s = L; s < SHINE_MAX_LEVELS; ++s) a.lv[s] = a.lv[0];  // keeps the unguarded speculative loads in bounds
  a.coord = coord;
  a.n = n;
  a.n_levels = L;
  a.res_leaf = (float)(1u << cfg->max_level);
  a.first_miss_bucket = (int)(t->n_buckets - MISS_BUCKETS);
  a.count = (int*)(w + o_count);
  a.bucket = (int*)(w + o_bucket);
  a.local = (int*)(w + o_local);
  a.zero_ptr = (float4*)zero_ptr;
  a.zero_n16 = zero_ptr ? (long long)(zero_bytes / 16) : 0;
  a.ablate = cfg->kernel_variant >> 8;
  a.slots_tmp = (int*)(w + o_slots);
  a.perm = (int*)perm_out;
  a.slots_sorted = (int*)slots_out;
  if (n <= PLAN_UNSORTED_MAX && !(a.ablate & 15)) {  // (kernel_variant 0x800: tests / measurement — the counting sort for every size)
    a.slots_tmp = a.slots_sorted;
    const long long pb = (n + 255) / 256, zb = (a.zero_n16 + 256 * 8 - 1) / (256 * 8);  // zeroing: ~8 float4 per thread
    hipLaunchKernelGGL(k_plan_unsorted, dim3((unsigned)(pb > zb ? pb : zb < 2048 ? zb : 2048)), dim3(256), 0, st, a);
    SHINE_HIP_CHECK(hipGetLastError());
    return SHINE_OK;
  }
  SHINE_HIP_CHECK(hipMemsetAsync(a.count, 0, nb * 4, st));
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_plan_count, grid, block, 0, st, a);
  SHINE_HIP_CHECK(hipGetLastError());
  SHINE_HIP_CHECK(prim_scan_int(w + o_scan, scan_bytes, a.count, a.count, nb, st));
  hipLaunchKernelGGL(k_plan_scatter, grid, block, 0, st, a);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}
