// kernarg_latency.hip — how long does a wave wait for its kernel arguments?  (the fused step's first 7 k cycles: lab book block 12)
// A kernel with a 560-byte by-value argument block stamps the shader clock at entry, after the first use of an argument
// (s_load + s_waitcnt), after a dependent global load through a pointer taken from the arguments, and after a 64-bit division by
// a run-time value (the tile split).  Plain launches and launches from a HIP graph; 64 and 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Big { long long v[64]; const int* p; long long n; long long* out; long long pad[3]; };  // 560 bytes

__global__ void k_big(Big a) {
  const long long t0 = clock64();
  long long s = a.v[threadIdx.x & 63];  // first use of the arguments
  asm volatile("" : "+v"(s));
  const long long t1 = clock64();
  int x = a.p[(blockIdx.x * 64 + (s & 31)) & 1023];  // a dependent global load
  asm volatile("" : "+v"(x));
  const long long t2 = clock64();
  long long q = ((long long)blockIdx.x * a.n) / (long long)gridDim.x;  // the tile split's division
  asm volatile("" : "+s"(q));
  const long long t3 = clock64();
  if ((threadIdx.x & 63) == 0) {
    long long* o = a.out + ((long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4;
    o[0] = t1 - t0, o[1] = t2 - t1, o[2] = t3 - t2, o[3] = q + x;
  }
}

int main() {
  int* p; long long* out;
  CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
  CK(hipMalloc(&out, 1 << 20));
  Big a = {}; a.p = p; a.n = 16384; a.out = out;
  for (int i = 0; i < 64; ++i) a.v[i] = i;
  hipStream_t st; CK(hipStreamCreate(&st));
  for (int blocks : {64, 256}) {
    for (int mode = 0; mode < 2; ++mode) {
      hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
      if (mode == 1) {
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int r = 0; r < 20; ++r) k_big<<<blocks, 256, 0, st>>>(a);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      }
      double acc[3] = {0, 0, 0}; long long mx[3] = {0, 0, 0}; int cnt = 0;
      for (int rep = 0; rep < 6; ++rep) {
        if (mode == 0) for (int r = 0; r < 20; ++r) k_big<<<blocks, 256, 0, st>>>(a);
        else CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        if (rep == 0) continue;  // warm-up
        std::vector<long long> h((size_t)blocks * 4 * 4);
        CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        for (int w = 0; w < blocks * 4; ++w) {
          for (int k = 0; k < 3; ++k) { acc[k] += (double)h[w * 4 + k]; if (h[w * 4 + k] > mx[k]) mx[k] = h[w * 4 + k]; }
          ++cnt;
        }
      }
      printf("%3d workgroups x 4 waves, %s: clock64 ticks per wave, mean (max): first use of the arguments %.0f (%lld), dependent global load %.0f (%lld), 64-bit division %.0f (%lld)\n",
             blocks, mode ? "graph of 20 launches" : "plain launches     ", acc[0] / cnt, mx[0], acc[1] / cnt, mx[1], acc[2] / cnt, mx[2]);
    }
  }
  // clock64 rate: ticks per microsecond
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  return 0;
}
