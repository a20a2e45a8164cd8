// grid_barrier.hip — what a device-wide barrier costs on MI355X (8 XCDs, one L2 each), the design input of the persistent
// iteration kernel (shine_iterate.hip): K barriers between G workgroups of T threads, each barrier = { __syncthreads, thread 0:
// release fence (agent scope: L2 write-back), atomic arrive, bounded spin on the counter, acquire fence (agent scope: L1 / L2
// invalidate), __syncthreads }.  Between barriers every workgroup stores a value with PLAIN stores and reads its neighbour's with
// PLAIN loads after the barrier: a stale read (another XCD's L2) is counted — the fences must make plain memory coherent.
//
//   hipcc --offload-arch=gfx950 -O3 -o bin/grid_barrier grid_barrier.hip ;  bin/grid_barrier
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e = (x);                                                       \
    if (e != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned int* ctr, unsigned int target, int sleep) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long long spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (sleep) __builtin_amdgcn_s_sleep(1);
      if (++spins > (1ll << 24)) {  // never hang the box: give up (reported as a failure)
        ok = false;
        break;
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (agent scope is the default of the builtin fence in HIP device code)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

// the same barrier as a two-level tree: workgroup b arrives at the counter of group b % 8 (round-robin dispatch: its XCD), the
// last arriver of a group arrives at the top counter, the last of those publishes the generation every workgroup spins on —
// 256 same-address atomics serialise (~25-40 ns each), 32 + 8 do not.  Counters are monotonic (no reset), 128 B apart.
__device__ __forceinline__ bool tree_barrier(unsigned int* st, unsigned int gen, int sleep) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned int G = gridDim.x, g = blockIdx.x & 7u, ng = G < 8u ? G : 8u;
    const unsigned int gs = (G - g + 7u) / 8u;  // workgroups of this group
    unsigned int* grp = st + 32u * (1u + g);
    unsigned int* top = st + 32u * 9u;
    unsigned int* pub = st;
    const unsigned int old = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (old == gen * gs - 1u) {
      const unsigned int old2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (old2 == gen * ng - 1u) __hip_atomic_store(pub, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    long long spins = 0;
    while (__hip_atomic_load(pub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen) {
      if (sleep) __builtin_amdgcn_s_sleep(1);
      if (++spins > (1ll << 24)) {
        ok = false;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

// mode 0: barrier only; mode 1: + each workgroup writes `bytes` of plain stores before the barrier and reads the neighbour's
template <bool TREE>
__global__ void k_barriers(unsigned int* ctr, int iters, float* data, int words_per_block, int* stale, int* gaveup, int sleep) {
  const int G = gridDim.x, b = blockIdx.x;
  int bad = 0;
  for (int it = 0; it < iters; ++it) {
    for (int w = threadIdx.x; w < words_per_block; w += blockDim.x) data[(long long)b * words_per_block + w] = (float)(it + 1);
    if (!(TREE ? tree_barrier(ctr, (unsigned)(it * 2 + 1), sleep) : grid_barrier(ctr, (unsigned)(it * 2 + 1) * G, sleep))) {
      if (threadIdx.x == 0) atomicAdd(gaveup, 1);
      return;
    }
    const int nb = (b + G / 2 + 1) % G;  // a workgroup far away (another XCD under round-robin dispatch)
    for (int w = threadIdx.x; w < words_per_block; w += blockDim.x)
      if (data[(long long)nb * words_per_block + w] != (float)(it + 1)) ++bad;
    // second barrier: nobody overwrites before everybody has read (the iteration kernel has two per iteration as well)
    if (!(TREE ? tree_barrier(ctr, (unsigned)(it * 2 + 2), sleep) : grid_barrier(ctr, (unsigned)(it * 2 + 2) * G, sleep))) {
      if (threadIdx.x == 0) atomicAdd(gaveup, 1);
      return;
    }
  }
  if (bad) atomicAdd(stale, bad);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  unsigned int* ctr;
  float* data;
  int *stale, *gaveup;
  CK(hipMalloc(&ctr, 4096));
  CK(hipMalloc(&data, 256ll * 65536 * 4));
  CK(hipMalloc(&stale, 4));
  CK(hipMalloc(&gaveup, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int grids[] = {64, 128, 256};
  const int threads[] = {64, 256};
  const int words[] = {0, 256, 16384};
  for (int tree = 0; tree < 2; ++tree)
  for (int sleep = 0; sleep < 2; ++sleep)
    for (int G : grids)
      for (int T : threads)
        for (int W : words) {
          float best = 1e30f;
          int st = 0, gu = 0;
          for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(ctr, 0, 4096));
            CK(hipMemset(stale, 0, 4));
            CK(hipMemset(gaveup, 0, 4));
            CK(hipEventRecord(e0));
            if (tree) hipLaunchKernelGGL(k_barriers<true>, dim3(G), dim3(T), 0, 0, ctr, iters, data, W, stale, gaveup, sleep);
            else hipLaunchKernelGGL(k_barriers<false>, dim3(G), dim3(T), 0, 0, ctr, iters, data, W, stale, gaveup, sleep);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            int s, g;
            CK(hipMemcpy(&s, stale, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&g, gaveup, 4, hipMemcpyDeviceToHost));
            st += s;
            gu += g;
          }
          printf("%s sleep %d grid %3d x %3d threads, %6d words/block between barriers: %.2f us per barrier (%d iterations x 2), stale reads %d, gave up %d\n",
                 tree ? "tree" : "flat", sleep, G, T, W, best * 1e3f / (2 * iters), iters, st, gu);
          fflush(stdout);
        }
  return 0;
}
