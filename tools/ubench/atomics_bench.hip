// atomics_bench.hip — what do global atomics cost on gfx950?  (design input for shine_plan.hip / the scatter)
//   N random addresses out of M counters; variants: agent vs workgroup scope, returning vs fire-and-forget, u32 vs f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void k(const int* idx, unsigned* cnt, float* fcnt, int* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int j = idx[i];
  if (MODE == 0) { out[i] = atomicAdd(&cnt[j], 1u); }                                                   // agent, returning
  if (MODE == 1) { atomicAdd(&cnt[j], 1u); }                                                            // agent, no return
  if (MODE == 2) { out[i] = __hip_atomic_fetch_add(&cnt[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  if (MODE == 3) { __hip_atomic_fetch_add(&cnt[j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  if (MODE == 4) { unsafeAtomicAdd(&fcnt[j], 1.0f); }                                                   // f32 hw atomic
  if (MODE == 5) { __hip_atomic_fetch_add(&fcnt[j], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
  if (MODE == 6) { out[i] = cnt[j]; }                                                                   // plain gather
  if (MODE == 7) { cnt[j] = 1u; }                                                                       // plain scatter
  if (MODE == 8) {  // 8 x 32-B rows per wave-instruction (the fused kernel's scatter shape): lane = (row, feature)
    int lane = threadIdx.x & 63; int row = idx[(i & ~63) + (lane >> 3)];
    unsafeAtomicAdd(&fcnt[(long long)row * 8 + (lane & 7)], 1.0f);
  }
}

template <int MODE>
float run(const int* idx, unsigned* cnt, float* fcnt, int* out, int n, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k<MODE><<<(n + 255) / 256, 256>>>(idx, cnt, fcnt, out, n);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; ++r) k<MODE><<<(n + 255) / 256, 256>>>(idx, cnt, fcnt, out, n);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1000.f / reps;
}

int main() {
  const int n = 1 << 18;
  const char* names[] = {"u32 agent returning", "u32 agent no-return", "u32 workgroup returning", "u32 workgroup no-return",
                         "f32 unsafeAtomicAdd", "f32 workgroup fetch_add", "plain gather (4 B)", "plain scatter (4 B)",
                         "f32 atomics, 8 rows x 32 B per wave-instr"};
  for (int m : {1 << 12, 1 << 18, 1 << 22}) {
    std::vector<int> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (int)(((long long)rand() * 65536 + rand()) % m);
    int *idx, *out; unsigned* cnt; float* fcnt;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&cnt, (size_t)m * 4)); CK(hipMalloc(&fcnt, (size_t)m * 32));
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipMemset(cnt, 0, (size_t)m * 4)); CK(hipMemset(fcnt, 0, (size_t)m * 32));
    printf("N = %d lane-ops on M = %d random targets\n", n, m);
    float t[9];
    t[0] = run<0>(idx, cnt, fcnt, out, n, 20); t[1] = run<1>(idx, cnt, fcnt, out, n, 20);
    t[2] = run<2>(idx, cnt, fcnt, out, n, 20); t[3] = run<3>(idx, cnt, fcnt, out, n, 20);
    t[4] = run<4>(idx, cnt, fcnt, out, n, 20); t[5] = run<5>(idx, cnt, fcnt, out, n, 20);
    t[6] = run<6>(idx, cnt, fcnt, out, n, 20); t[7] = run<7>(idx, cnt, fcnt, out, n, 20);
    t[8] = run<8>(idx, cnt, fcnt, out, n, 20);
    for (int v = 0; v < 9; ++v) printf("   %-44s %8.2f us  (%6.2f G lane-ops/s)\n", names[v], t[v], n / t[v] / 1e3);
    CK(hipFree(idx)); CK(hipFree(out)); CK(hipFree(cnt)); CK(hipFree(fcnt));
  }
  return 0;
}
