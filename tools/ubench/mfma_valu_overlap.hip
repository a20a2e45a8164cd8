// Does exact-fp32 MFMA (v_mfma_f32_16x16x4_f32) overlap with fp32 VALU work of ANOTHER wave on the same SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/mfma_valu_overlap tools/ubench/mfma_valu_overlap.hip
// Launch: 256 workgroups x 512 threads (8 waves per CU = 2 per SIMD).  Waves wv = 0..3 and wv = 4..7 land on the same four
// SIMDs (dispatch order 0,2,1,3 cyclic), so wave wv and wave wv + 4 share a SIMD.
//   mode 0: waves 0-3 issue MFMAs, waves 4-7 exit        mode 1: waves 0-3 exit, waves 4-7 issue v_fma_f32
//   mode 2: both                                          mode 3: all 8 waves MFMA       mode 4: all 8 waves VALU
//   modes 1 / 2 / 4 issue v_pk_fma_f32 (hipcc SLP-packs the adjacent fmas: 2 fma per instruction).  Modes 5 / 6 / 7 are
//   modes 1 / 4 / 2 with PLAIN v_fma_f32 (inline asm, one fma per instruction) — what most of the fused step's vector
//   instructions are; they calibrate the cycles one plain VALU wave-instruction takes out of the fp32 datapath.
// If the matrix pipe and the vector ALU were independent units, t(2) ~ max(t(0), t(1)); if fp32 MFMA runs on the vector
// ALU's FMA lanes, t(2) ~ t(0) + t(1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void k(int mode, int iters, float* out) {
  const int wv = threadIdx.x >> 6;
  const bool do_mfma = (mode == 0 && wv < 4) || (mode == 2 && wv < 4) || mode == 3;
  const bool do_valu = (mode == 1 && wv >= 4) || (mode == 2 && wv >= 4) || mode == 4;
  const bool do_plain = (mode == 5 && wv >= 4) || mode == 6 || (mode == 7 && wv >= 4);
  const bool do_mfma7 = mode == 7 && wv < 4;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  if (do_plain) {
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = a + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(b), "v"(a));
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    return;
  }
  if (do_mfma || do_mfma7) {
    f32x4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += c[j][0] + c[j][1] + c[j][2] + c[j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else if (do_valu) {
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = a + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], b, a);  // 16 independent v_fma_f32 per iteration
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], b, a);
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], b, a);
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], b, a);
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;  // one mode per process: the PMC calibration runs (tools/gpu_calibrate.sh)
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;  // per wave: 8 x 4000 MFMAs (32 cycles each) or 64 x 4000 v_fma
  for (int rep = 0; rep < 2; ++rep)
    for (int mode = 0; mode < 8; ++mode) {
      if (only >= 0 && mode != only) continue;
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, iters, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      const double mfma_cycles = 8.0 * iters * 32, valu_instr = 64.0 * iters;
      if (rep)
        printf("mode %d: %8.3f ms   (one wave's work: %.0f MFMA-issue cycles, %.0f fma = %.0f %s)\n", mode, ms, mfma_cycles,
               valu_instr, mode >= 5 ? valu_instr : valu_instr / 2, mode >= 5 ? "v_fma_f32" : "v_pk_fma_f32");
    }
  return 0;
}
