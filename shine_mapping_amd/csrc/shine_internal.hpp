// shine_internal.hpp — host-side state shared by the translation units of libshine_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <new>
#include <vector>

#include "shine_device.hpp"

namespace shine {

struct TableLevel {
  unsigned long long* keys = nullptr;
  int* vals = nullptr;
  int* ranks = nullptr;  // [cap] bucket rank of the node in the all-level Z-order (shine_tables_set_ranks), or null
  long long cap = 0;
  long long count = 0;
  unsigned int shift = 0;
  unsigned int mask = 0;
};

// corners_lookup_tables[level] (model/feature_octree.py:47-52,135-152): corner (x,y,z) -> feature row id.
// Open addressing like TableLevel; key = x<<42 | y<<21 | z (ascending = the lexicographic order of
// torch.unique(dim=0), :132).  Only shine_tables_grow reads it (octree growth, once per frame).
struct CornerLevel {
  unsigned long long* keys = nullptr;
  int* vals = nullptr;
  long long cap = 0;
  long long count = 0;
  unsigned int shift = 0;
  unsigned int mask = 0;
};

// device scratch of one shine_tables_grow call; results stay readable for shine_tables_grow_fetch
struct GrowScratch {
  void* a = nullptr;  // phase A: leaf keys, flags, fresh node keys
  size_t a_bytes = 0;
  void* b = nullptr;  // phase B/C: corner keys, new corners, ids
  size_t b_bytes = 0;
  long long n_fresh[SHINE_MAX_LEVELS] = {};
  long long n_added[SHINE_MAX_LEVELS] = {};
  unsigned long long* fresh_keys[SHINE_MAX_LEVELS] = {};  // [n_fresh] Morton order
  int* fresh_ids[SHINE_MAX_LEVELS] = {};                  // [n_fresh][8]
  unsigned long long* new_corners[SHINE_MAX_LEVELS] = {}; // [n_added] lexicographic = id order
};

// One optimiser step counted in a shine_adam_step_dev state (int64[8]: [0] steps taken, [1] the two bias corrections as
// floats, [2] / [3] beta1^t / beta2^t as doubles).  The powers are RUNNING PRODUCTS — one multiply per step; a double pow()
// is ~10 k cycles of one thread — and 0 means "not initialised": derived with pow once.  Called by ONE thread per step:
// k_adam_prep (shine_adam.hip) or the step's reduction launch (k_reduce_partials, cfg->adam_state).
__device__ inline void adam_advance(long long* state, float b1, float b2) {
  const long long t = state[0] + 1;
  state[0] = t;
  double* pr = reinterpret_cast<double*>(state + 2);
  double p1 = pr[0], p2 = pr[1];
  if (p1 == 0.0 || p2 == 0.0) {
    p1 = pow((double)b1, (double)(t - 1));
    p2 = pow((double)b2, (double)(t - 1));
  }
  p1 *= (double)b1;
  p2 *= (double)b2;
  pr[0] = p1;
  pr[1] = p2;
  float* bc = reinterpret_cast<float*>(state + 1);
  bc[0] = (float)(1.0 - p1);
  bc[1] = (float)sqrt(1.0 - p2);
}

int set_error(int code, const char* msg);
int set_hip_error(hipError_t e, const char* what);

}  // namespace shine

struct shine_tables {
  // Device arrays a growth replaced (a rehashed table, an outgrown scratch buffer): kept until the handle is destroyed instead
  // of being freed on the spot.  hipFree waits for the whole device and was measured at 40-68 ms inside an incremental run whose
  // frames take 3.4 ms; the arrays double, so what is retired never exceeds what is live.  shine_tables_trim frees them once
  // the caller knows that no launch bound to them is pending.
  std::vector<void*> retired;
  int n_levels = 0;
  long long n_buckets = 0;  // nodes of all featured levels + 1 ("misses everywhere"); 0: ranks not set
  shine::TableLevel lv[SHINE_MAX_LEVELS];
  shine::CornerLevel cl[SHINE_MAX_LEVELS];
  shine::GrowScratch grow;
  long long retired_bytes = 0;
  void retire(void* p, size_t bytes) {
    if (!p) return;
    retired.push_back(p);
    retired_bytes += (long long)bytes;
  }
};

#define SHINE_HIP_CHECK(expr)                                      \
  do {                                                             \
    hipError_t e__ = (expr);                                       \
    if (e__ != hipSuccess) return shine::set_hip_error(e__, #expr); \
  } while (0)

namespace shine {

// Fill the kernel-side level descriptors from the handle + the caller's feature/grad pointers.
// Returns SHINE_OK or an error code (message set).
inline int make_level_set(const shine_tables* t, const shine_step_config* cfg, const float* const* feats,
                          const int64_t* rows, float* const* grads, LevelSet* out) {
  if (!t || !cfg) return set_error(SHINE_E_INVALID, "null tables/config");
  const int L = cfg->n_levels;
  if (L < 1 || L > SHINE_MAX_LEVELS || L != t->n_levels) return set_error(SHINE_E_INVALID, "n_levels mismatch");
  if (cfg->max_level < L || cfg->max_level > 15) return set_error(SHINE_E_INVALID, "max_level out of range (<=15)");
  for (int s = 0; s < L; ++s) {
    const TableLevel& T = t->lv[s];
    if (!T.keys) return set_error(SHINE_E_STATE, "a featured level has no table yet (call shine_tables_insert)");
    LevelDev& D = out->lv[s];
    D.keys = T.keys;
    D.vals = reinterpret_cast<const int4*>(T.vals);
    D.feat = feats ? feats[s] : nullptr;
    D.grad = grads ? grads[s] : nullptr;
    D.rows = rows ? rows[s] : 0;
    D.shift = T.shift;
    D.mask = T.mask;
    const int level = cfg->max_level - (L - 1 - s);
    D.res = (float)(1u << level);
    D.dres = D.res * 0.5f;
  }
  return SHINE_OK;
}

// rocPRIM behind four host wrappers, instantiated once (shine_prims.hip).  tmp == nullptr: `bytes` = temporary storage needed.
hipError_t prim_scan_int(void* tmp, size_t& bytes, const int* in, int* out, size_t n, hipStream_t st);
hipError_t prim_scan_flags(void* tmp, size_t& bytes, const unsigned char* flags, int* out, size_t n, hipStream_t st);
hipError_t prim_sort_keys_u64(void* tmp, size_t& bytes, const unsigned long long* k0, unsigned long long* k1, size_t n,
                              unsigned begin_bit, unsigned end_bit, hipStream_t st);
hipError_t prim_sort_segments_4k(int* idx, long long seg, long long total, int n_seg, unsigned end_bit, hipStream_t st);
hipError_t prim_warmup(hipStream_t st);  // once per process: loads rocPRIM's size-class kernels (shine_prims.hip)
hipError_t prim_sort_pairs_u64(void* tmp, size_t& bytes, const unsigned long long* k0, unsigned long long* k1,
                               const unsigned long long* v0, unsigned long long* v1, size_t n, unsigned begin_bit,
                               unsigned end_bit, hipStream_t st);

}  // namespace shine
