// atomics_rows.hip — the L2's fp32 atomic rate for the fused step's scatter shape (profiles/r04_ab_experiments.txt block 10):
// W wave-instructions of global_atomic_add_f32, each 8 rows x 32 B (lane = (row, feature)), rows drawn
//   (a) uniformly from M rows, (b) as 8 CONSECUTIVE-ish rows (corner ids of one node are allocated close together),
// nothing else in the kernel: what the split step's second launch can reach at best.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_rows(const int* __restrict__ rows, float* __restrict__ table, int wave_ops, int per_wave) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  for (int j = 0; j < per_wave; ++j) {
    const long long op = wave * per_wave + j;
    if (op >= wave_ops) return;
    const int row = rows[op * 8 + (lane >> 3)];
    unsafeAtomicAdd(&table[(long long)row * 8 + (lane & 7)], 1.0f);
  }
}

// Round 6: the same 8 rows x 32 B per wave instruction as 64-bit atomics — lane = (row, feature PAIR), 32 active lanes: is the
// memory side's rate per lane (then wider atomics halve the time for the same payload) or per byte?  WIDE 1: u64 integer add
// (two fixed-point halves could share one), 2: f64 add.
template <int WIDE>
__global__ void k_rows_wide(const int* __restrict__ rows, float* __restrict__ table, int wave_ops, int per_wave) {
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  for (int j = 0; j < per_wave; ++j) {
    const long long op = wave * per_wave + j;
    if (op >= wave_ops) return;
    if (lane >= 32) continue;
    const int row = rows[op * 8 + (lane >> 2)];
    void* p = &table[(long long)row * 8 + 2 * (lane & 3)];
    if (WIDE == 1) atomicAdd(reinterpret_cast<unsigned long long*>(p), 0x0000000100000001ull);
    else unsafeAtomicAdd(reinterpret_cast<double*>(p), 1.0);
  }
}

int main() {
  const int wave_ops = 131072;  // the headline step: 8.0 node runs per 16-point tile x 16384 tiles
  for (long long M : {400000ll, 4000000ll}) {
    for (int mode = 0; mode < 3; ++mode) {
      std::vector<int> h((size_t)wave_ops * 8);
      srand(7);
      for (int w = 0; w < wave_ops; ++w) {
        const long long base = ((long long)rand() * 65536 + rand()) % (M - 64);
        // mode 2: consecutive wave-ops hit NEIGHBOURING nodes (shared corners): the base advances by ~2 rows per op
        const long long b2 = ((long long)w * 3) % (M - 64);
        for (int c = 0; c < 8; ++c)
          h[(size_t)w * 8 + c] = mode == 0 ? (int)(((long long)rand() * 65536 + rand()) % M)
                               : mode == 1 ? (int)(base + (rand() % 32)) : (int)(b2 + (rand() % 16));
      }
      int* rows; float* table;
      CK(hipMalloc(&rows, h.size() * 4)); CK(hipMalloc(&table, (size_t)M * 32));
      CK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemset(table, 0, (size_t)M * 32));
      for (int per_wave : {1, 8, 32}) {
        const int waves = (wave_ops + per_wave - 1) / per_wave;
        const int blocks = (waves * 64 + 255) / 256;
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        k_rows<<<blocks, 256>>>(rows, table, wave_ops, per_wave);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int r = 0; r < 20; ++r) k_rows<<<blocks, 256>>>(rows, table, wave_ops, per_wave);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double us = ms * 1000.0 / 20;
        printf("M = %lld rows, %s, %d ops per wave: %.1f us for %d wave-atomics (8 rows x 32 B each) = %.1f row-atomics/ns, %.2f fp32 atomics per ns\n",
               M, mode == 0 ? "rows uniform" : mode == 1 ? "8 rows within 32 of a random base" : "neighbouring bases (3 rows apart)",
               per_wave, us, wave_ops, wave_ops * 8 / us / 1e3, wave_ops * 64 / us / 1e3);
      }
      for (int wide = 1; wide <= 2; ++wide) {
        const int per_wave = 8;
        const int waves = (wave_ops + per_wave - 1) / per_wave;
        const int blocks = (waves * 64 + 255) / 256;
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        auto launch = [&]() {
          if (wide == 1) k_rows_wide<1><<<blocks, 256>>>(rows, table, wave_ops, per_wave);
          else k_rows_wide<2><<<blocks, 256>>>(rows, table, wave_ops, per_wave);
        };
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int r = 0; r < 20; ++r) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double us = ms * 1000.0 / 20;
        printf("M = %lld rows, mode %d, 8 ops per wave, 64-BIT %s atomics (32 lanes per instruction, the same 8 rows x 32 B): %.1f us = %.1f row-atomics/ns, %.2f lane atomics per ns\n",
               M, mode, wide == 1 ? "u64 add" : "f64 add", us, wave_ops * 8 / us / 1e3, wave_ops * 32 / us / 1e3);
      }
      CK(hipFree(rows)); CK(hipFree(table));
    }
  }
  return 0;
}
