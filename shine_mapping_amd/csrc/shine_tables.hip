// shine_tables.hip — device hash tables that replace the reference's per-level Python dicts
// nodes_lookup_tables[level] : node Morton -> [8 corner ids]   (model/feature_octree.py:47-52,162-166,209).
//
// Layout per featured level: open addressing, linear probing, capacity a power of two >= 2 x nodes.
//   keys[cap]      uint64  Morton code (kaolin convention) or EMPTY_KEY
//   vals[cap][8]   int32   corner (feature-row) ids, corner order of interpolat (:186-193)
// keys and vals share the slot index, so a probe issues the key load and the 32-B id load together and
// the dependent chain of a query is {probe} -> {8 row gathers}.  Inserts run on the device
// (atomicCAS claim), growth rehashes on the device; both happen per frame (FeatureOctree.update), never
// inside a training iteration.
#include "shine_internal.hpp"

namespace shine {

__global__ void k_fill_keys(unsigned long long* keys, long long cap) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) keys[i] = EMPTY_KEY;
}

__global__ void k_insert(unsigned long long* keys, int* vals, unsigned int shift, unsigned int mask,
                         const long long* in_keys, const int* in_vals, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long key = (unsigned long long)in_keys[i];
  unsigned int slot = hash_slot(key, shift);
  for (unsigned int probe_n = 0; probe_n <= mask; ++probe_n) {
    unsigned long long prev = atomicCAS(&keys[slot], EMPTY_KEY, key);
    if (prev == EMPTY_KEY || prev == key) {
      const int4* src = reinterpret_cast<const int4*>(in_vals + i * 8);
      int4* dst = reinterpret_cast<int4*>(vals + (long long)slot * 8);
      dst[0] = src[0];
      dst[1] = src[1];
      return;
    }
    slot = (slot + 1) & mask;
  }
}

__global__ void k_rehash(const unsigned long long* old_keys, const int* old_vals, long long old_cap,
                         unsigned long long* keys, int* vals, unsigned int shift, unsigned int mask) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= old_cap) return;
  unsigned long long key = old_keys[i];
  if (key == EMPTY_KEY) return;
  unsigned int slot = hash_slot(key, shift);
  for (unsigned int probe_n = 0; probe_n <= mask; ++probe_n) {
    unsigned long long prev = atomicCAS(&keys[slot], EMPTY_KEY, key);
    if (prev == EMPTY_KEY) {
      const int4* src = reinterpret_cast<const int4*>(old_vals + i * 8);
      int4* dst = reinterpret_cast<int4*>(vals + (long long)slot * 8);
      dst[0] = src[0];
      dst[1] = src[1];
      return;
    }
    slot = (slot + 1) & mask;
  }
}

__global__ void k_set_ranks(const unsigned long long* keys, int* ranks, unsigned int shift, unsigned int mask,
                            const long long* in_keys, const int* in_ranks, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned long long key = (unsigned long long)in_keys[i];
  unsigned int slot = hash_slot(key, shift);
  for (unsigned int probe_n = 0; probe_n <= mask; ++probe_n) {
    const unsigned long long k = keys[slot];
    if (k == key) {
      ranks[slot] = in_ranks[i];
      return;
    }
    if (k == EMPTY_KEY) return;
    slot = (slot + 1) & mask;
  }
}

static int alloc_level(TableLevel& L, long long cap, hipStream_t st) {
  unsigned long long* keys = nullptr;
  int* vals = nullptr;
  int* ranks = nullptr;
  if (hipMalloc(&keys, (size_t)cap * sizeof(unsigned long long)) != hipSuccess) return SHINE_E_NOMEM;
  if (hipMalloc(&vals, (size_t)cap * 8 * sizeof(int)) != hipSuccess) {
    (void)hipFree(keys);
    return SHINE_E_NOMEM;
  }
  if (hipMalloc(&ranks, (size_t)cap * sizeof(int)) != hipSuccess) {
    (void)hipFree(keys);
    (void)hipFree(vals);
    return SHINE_E_NOMEM;
  }
  L.ranks = ranks;
  hipLaunchKernelGGL(k_fill_keys, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, keys, cap);
  SHINE_HIP_CHECK(hipGetLastError());
  L.keys = keys;
  L.vals = vals;
  L.cap = cap;
  int lg = 0;
  while ((1ll << lg) < cap) ++lg;
  L.shift = 64u - (unsigned)lg;
  L.mask = (unsigned)(cap - 1);
  return SHINE_OK;
}

}  // namespace shine

using namespace shine;

extern "C" int shine_tables_create(int32_t n_levels, shine_tables** out) {
  if (!out || n_levels < 1 || n_levels > SHINE_MAX_LEVELS) return set_error(SHINE_E_INVALID, "n_levels out of range");
  shine_tables* t = new (std::nothrow) shine_tables();
  if (!t) return set_error(SHINE_E_NOMEM, "host allocation failed");
  t->n_levels = n_levels;
  *out = t;
  return SHINE_OK;
}

extern "C" int shine_tables_destroy(shine_tables* t) {
  if (!t) return SHINE_OK;
  for (int s = 0; s < SHINE_MAX_LEVELS; ++s) {
    if (t->lv[s].keys) (void)hipFree(t->lv[s].keys);
    if (t->lv[s].vals) (void)hipFree(t->lv[s].vals);
    if (t->lv[s].ranks) (void)hipFree(t->lv[s].ranks);
    if (t->cl[s].keys) (void)hipFree(t->cl[s].keys);
    if (t->cl[s].vals) (void)hipFree(t->cl[s].vals);
  }
  if (t->grow.a) (void)hipFree(t->grow.a);
  if (t->grow.b) (void)hipFree(t->grow.b);
  for (void* p : t->retired) (void)hipFree(p);
  delete t;
  return SHINE_OK;
}

extern "C" int shine_tables_retired_bytes(const shine_tables* t, int64_t* bytes) {
  if (!t || !bytes) return set_error(SHINE_E_INVALID, "shine_tables_retired_bytes: null argument");
  *bytes = t->retired_bytes;
  return SHINE_OK;
}

extern "C" int shine_tables_trim(shine_tables* t, int64_t* freed_bytes) {
  if (!t) return set_error(SHINE_E_INVALID, "shine_tables_trim: null handle");
  for (void* p : t->retired) (void)hipFree(p);
  if (freed_bytes) *freed_bytes = t->retired_bytes;
  t->retired.clear();
  t->retired_bytes = 0;
  return SHINE_OK;
}

extern "C" int shine_tables_stats(const shine_tables* t, int32_t slot, int64_t* capacity, int64_t* count) {
  if (!t || slot < 0 || slot >= t->n_levels) return set_error(SHINE_E_INVALID, "bad table/slot");
  if (capacity) *capacity = t->lv[slot].cap;
  if (count) *count = t->lv[slot].count;
  return SHINE_OK;
}

extern "C" int shine_tables_insert(shine_tables* t, int32_t slot, const int64_t* keys, const int32_t* corner_ids,
                                   int64_t n, void* stream) {
  if (!t || slot < 0 || slot >= t->n_levels || n < 0) return set_error(SHINE_E_INVALID, "bad table/slot/n");
  if (n == 0) return SHINE_OK;
  if (!keys || !corner_ids) return set_error(SHINE_E_INVALID, "null keys/corner_ids");
  hipStream_t st = (hipStream_t)stream;
  TableLevel& L = t->lv[slot];
  long long need = L.count + n;
  long long cap = L.cap ? L.cap : 1024;
  while (cap < 2 * need) cap <<= 1;
  if (cap > (1ll << 31)) return set_error(SHINE_E_INVALID, "level exceeds 2^31 hash slots");
  if (cap != L.cap) {
    TableLevel fresh;
    int rc = alloc_level(fresh, cap, st);
    if (rc != SHINE_OK) return set_error(rc, "hash table allocation failed");
    if (L.cap) {
      hipLaunchKernelGGL(k_rehash, dim3((unsigned)((L.cap + 255) / 256)), dim3(256), 0, st, L.keys, L.vals,
                         (long long)L.cap, fresh.keys, fresh.vals, fresh.shift, fresh.mask);
      SHINE_HIP_CHECK(hipGetLastError());
      t->retire(L.keys, (size_t)L.cap * sizeof(unsigned long long));  // (no free on the spot: shine_internal.hpp)
      t->retire(L.vals, (size_t)L.cap * 8 * sizeof(int));
      t->retire(L.ranks, (size_t)L.cap * sizeof(int));
    }
    fresh.count = L.count;
    L = fresh;
  }
  t->n_buckets = 0;  // node ranks are stale after any insert: shine_tables_set_ranks must run again
  hipLaunchKernelGGL(k_insert, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, L.keys, L.vals, L.shift, L.mask,
                     (const long long*)keys, (const int*)corner_ids, (long long)n);
  SHINE_HIP_CHECK(hipGetLastError());
  L.count += n;
  return SHINE_OK;
}

extern "C" int shine_tables_set_ranks(shine_tables* t, int32_t slot, const int64_t* keys, const int32_t* ranks,
                                      int64_t n, int64_t n_buckets, void* stream) {
  if (!t || slot < 0 || slot >= t->n_levels || n < 0 || n_buckets < 1)
    return set_error(SHINE_E_INVALID, "shine_tables_set_ranks: bad table/slot/n");
  TableLevel& L = t->lv[slot];
  if (!L.keys) return set_error(SHINE_E_STATE, "shine_tables_set_ranks: level has no table yet");
  if (n > 0) {
    if (!keys || !ranks) return set_error(SHINE_E_INVALID, "shine_tables_set_ranks: null keys/ranks");
    hipLaunchKernelGGL(k_set_ranks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, L.keys,
                       L.ranks, L.shift, L.mask, (const long long*)keys, (const int*)ranks, (long long)n);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  if (slot == t->n_levels - 1) t->n_buckets = n_buckets;  // by convention the leaf level is set last
  return SHINE_OK;
}
