// shine_grow.hip — octree growth on the device: FeatureOctree.update (model/feature_octree.py:114-166).
//
// The reference walks every node and corner of a scan in Python (:125-127,149-151,163-166; ~9 s for a 400 k-point
// frame).  Here one call turns the frame's surface points into table inserts without the points or the tables
// leaving HBM.  What must come out bit-identical (it fixes which feature row every corner owns):
//   * nodes of a level = unique(quantised leaf voxels >> 3(Lmax-l)), visited in Morton order (kaolin
//     unbatched_pointcloud_to_spc + point_hierarchies, :116-123); "new" = not yet in nodes_lookup_tables[l] (:125-127)
//   * their corners, deduplicated in LEXICOGRAPHIC (x,y,z) order (torch.unique(dim=0), :132); corners not yet in
//     corners_lookup_tables[l] get the next row ids in that order (:135-137 first frame, :148-151 later)
//   * every new node stores its 8 corner ids in the corner order of points_to_corners (:162-166)
//
// Pipeline (integer work, HBM/latency bound; rocPRIM radix sort + scan are the only library calls):
//   A  leaf keys (same fp32 quantise as the query)  -> radix sort -> per level: flag (first of its run && not in the
//      node table) -> scan -> compact                 => fresh node keys, Morton order            [1 host sync: counts]
//   B  per level: 8 corner keys per fresh node -> radix sort -> flag (first of run && not in the corner table) ->
//      scan -> compact                                => new corner keys, lexicographic order     [1 host sync: counts]
//   C  per level: insert new corners with ids base+i; look the 8 corners of every fresh node up; insert the nodes.
// The two syncs return 2L integers the host needs anyway (rows to append to the feature tables, :139,153).
#include <cstring>

#include "shine_internal.hpp"

namespace shine {

using u64 = unsigned long long;

static size_t galign(size_t v) { return (v + 255) & ~(size_t)255; }

__device__ __forceinline__ unsigned int compact3(u64 m) {
  u64 x = m & 0x1249249249249249ull;
  x = (x | (x >> 2)) & 0x10C30C30C30C30C3ull;
  x = (x | (x >> 4)) & 0x100F00F00F00F00Full;
  x = (x | (x >> 8)) & 0x001F0000FF0000FFull;
  x = (x | (x >> 16)) & 0x001F00000000FFFFull;
  x = (x | (x >> 32)) & 0xFFFFull;
  return (unsigned int)x;
}

__device__ __forceinline__ u64 corner_key(u64 node_morton, int c) {
  const u64 x = compact3(node_morton >> 2) + ((c >> 2) & 1), y = compact3(node_morton >> 1) + ((c >> 1) & 1),
            z = compact3(node_morton) + (c & 1);
  return (x << 42) | (y << 21) | z;
}

struct ProbeTable {  // either table kind, read-only
  const u64* keys;
  unsigned int shift, mask;
};
__device__ __forceinline__ int find_slot(const ProbeTable& T, u64 key) {
  if (!T.keys) return -1;
  unsigned int slot = hash_slot(key, T.shift);
  for (unsigned int n = 0; n <= T.mask; ++n) {
    const u64 k = T.keys[slot];
    if (k == key) return (int)slot;
    if (k == EMPTY_KEY) return -1;
    slot = (slot + 1) & T.mask;
  }
  return -1;
}

__global__ void k_leaf_keys(const float* pts, long long n, float res, u64* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = morton3(quantize(pts[3 * i], res), quantize(pts[3 * i + 1], res), quantize(pts[3 * i + 2], res));
}



// ---- the same three steps for ALL featured levels in one launch each (blockIdx.y = level): an octree update is a chain of
//      ~10^2 small launches whose cost is the host's launch rate, not the GPU's (tools/update_breakdown.py)
struct LevelProbes {
  ProbeTable T[SHINE_MAX_LEVELS];
  int sh[SHINE_MAX_LEVELS];
  u64* out[SHINE_MAX_LEVELS];
};
__global__ void k_flag_fresh_levels(const u64* sorted, long long n, LevelProbes P, int* flags) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = blockIdx.y, sh = P.sh[s];
  const u64 k = sorted[i] >> sh;
  const bool first = i == 0 || (sorted[i - 1] >> sh) != k;
  flags[(long long)s * n + i] = (first && find_slot(P.T[s], k) < 0) ? 1 : 0;
}
// pos = exclusive scan of the L x n flags: level s's fresh nodes go to out[s][pos - pos[s n]]
__global__ void k_compact_levels(const u64* src, long long n, LevelProbes P, const int* flags, const int* pos,
                                 long long* count_out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int s = blockIdx.y;
  const long long j = (long long)s * n + i;
  const int base = pos[(long long)s * n];
  if (flags[j]) P.out[s][pos[j] - base] = src[i] >> P.sh[s];
  if (i == n - 1) count_out[s] = (long long)(pos[j] + flags[j] - base);
}

// corners of the fresh nodes of all levels in ONE array, the level in the key's top three bits (x << 42 | y << 21 | z uses 58)
constexpr int CORNER_TAG_SHIFT = 61;
constexpr u64 CORNER_KEY_MASK = (1ull << CORNER_TAG_SHIFT) - 1ull;
struct ExpandLevels {
  const u64* fresh[SHINE_MAX_LEVELS];
  long long nf[SHINE_MAX_LEVELS];
  long long off[SHINE_MAX_LEVELS];  // first output element of the level (8 per fresh node)
};
__global__ void k_expand_corners_levels(ExpandLevels E, u64* out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (i >= E.nf[s] * 8) return;
  out[E.off[s] + i] = ((u64)s << CORNER_TAG_SHIFT) | corner_key(E.fresh[s][i >> 3], (int)(i & 7));
}
struct CornerProbes {
  ProbeTable T[SHINE_MAX_LEVELS];
};
// sorted by (level, lexicographic corner key): flag the first of every run whose corner table does not hold it
__global__ void k_flag_fresh_corners(const u64* sorted, long long m, CornerProbes P, int* flags) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const u64 k = sorted[i];
  const bool first = i == 0 || sorted[i - 1] != k;
  flags[i] = (first && find_slot(P.T[(int)(k >> CORNER_TAG_SHIFT)], k & CORNER_KEY_MASK) < 0) ? 1 : 0;
}
// new corners of all levels back to back (level by level, lexicographic inside a level); seg[l] = where level l starts
__global__ void k_compact_corners(const u64* sorted, long long m, int L, const int* flags, const int* pos, u64* out,
                                  long long* seg) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const u64 k = sorted[i];
  const int lvl = (int)(k >> CORNER_TAG_SHIFT);
  if (flags[i]) out[pos[i]] = k & CORNER_KEY_MASK;
  const int prev = i == 0 ? -1 : (int)(sorted[i - 1] >> CORNER_TAG_SHIFT);
  for (int l = prev + 1; l <= lvl; ++l) seg[l] = pos[i];  // (levels without fresh nodes in between start — and end — here)
  if (i == m - 1)
    for (int l = lvl + 1; l <= L; ++l) seg[l] = (long long)pos[i] + flags[i];
}


__global__ void k_fill_u64(u64* p, long long n, u64 v) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// ids: explicit array, or base + i when ids == nullptr
__global__ void k_insert_corners(u64* keys, int* vals, unsigned int shift, unsigned int mask, const u64* in_keys,
                                 const int* ids, int base, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 key = in_keys[i];
  unsigned int slot = hash_slot(key, shift);
  for (unsigned int probe_n = 0; probe_n <= mask; ++probe_n) {
    const u64 prev = atomicCAS(&keys[slot], EMPTY_KEY, key);
    if (prev == EMPTY_KEY || prev == key) {
      vals[slot] = ids ? ids[i] : base + (int)i;
      return;
    }
    slot = (slot + 1) & mask;
  }
}

__global__ void k_rehash_corners(const u64* old_keys, const int* old_vals, long long old_cap, u64* keys, int* vals,
                                 unsigned int shift, unsigned int mask) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= old_cap) return;
  const u64 key = old_keys[i];
  if (key == EMPTY_KEY) return;
  unsigned int slot = hash_slot(key, shift);
  for (unsigned int probe_n = 0; probe_n <= mask; ++probe_n) {
    if (atomicCAS(&keys[slot], EMPTY_KEY, key) == EMPTY_KEY) {
      vals[slot] = old_vals[i];
      return;
    }
    slot = (slot + 1) & mask;
  }
}

__global__ void k_corner_ids(const u64* fresh, long long nf, ProbeTable T, const int* vals, int* ids_out) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nf * 8) return;
  const int slot = find_slot(T, corner_key(fresh[i >> 3], (int)(i & 7)));
  ids_out[i] = slot >= 0 ? vals[slot] : -1;  // -1 cannot happen: phase C inserted every corner of every fresh node
}

// ---- node ranks on the device (shine_plan_batch / shine_sample_sorted order): every node of every level in ONE
//      Z-order, a parent's own bucket right after its children's.
__global__ void k_rank_keys(const u64* keys, long long cap, int sh, int depth_tag, int slot_tag, u64* out_keys,
                            u64* out_vals) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cap) return;
  const u64 k = keys[i];
  // end of the node's subtree range at leaf resolution, deeper levels first on ties
  out_keys[i] = (k == EMPTY_KEY) ? ~0ull : ((((k << sh) | ((1ull << sh) - 1ull)) << 3) | (u64)depth_tag);
  out_vals[i] = ((u64)slot_tag << 32) | (u64)i;
}
struct RankPtrs {
  int* ranks[SHINE_MAX_LEVELS];
};
__global__ void k_rank_scatter(const u64* sorted_vals, long long total_nodes, RankPtrs P) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total_nodes) return;
  const u64 v = sorted_vals[i];
  P.ranks[(int)(v >> 32)][(unsigned int)v] = (int)i;
}

static int ensure(shine_tables* t, void** ptr, size_t* have, size_t need, hipStream_t st) {
  (void)st;
  if (need <= *have) return SHINE_OK;
  if (*ptr) {
    t->retire(*ptr, *have);  // (launches that read it may still be in flight: no free, no sync — shine_internal.hpp)
    *ptr = nullptr;
    *have = 0;
  }
  // headroom so that the next frames do not outgrow it again: the need again, but at most 64 MiB (a 10^7-row map's
  // scratch is hundreds of MB, and doubling it held that much idle)
  const size_t want = need + (need < ((size_t)64 << 20) ? need : ((size_t)64 << 20));
  if (hipMalloc(ptr, want) != hipSuccess) return set_error(SHINE_E_NOMEM, "shine_tables_grow: scratch allocation failed");
  *have = want;
  return SHINE_OK;
}

static int corner_reserve(shine_tables* t, CornerLevel& Cn, long long need, hipStream_t st) {
  long long cap = Cn.cap ? Cn.cap : 1024;
  while (cap < 2 * need) cap <<= 1;
  if (cap == Cn.cap) return SHINE_OK;
  if (cap > (1ll << 31)) return set_error(SHINE_E_INVALID, "corner table exceeds 2^31 slots");
  u64* keys = nullptr;
  int* vals = nullptr;
  if (hipMalloc(&keys, (size_t)cap * 8) != hipSuccess) return set_error(SHINE_E_NOMEM, "corner table allocation failed");
  if (hipMalloc(&vals, (size_t)cap * 4) != hipSuccess) {
    (void)hipFree(keys);
    return set_error(SHINE_E_NOMEM, "corner table allocation failed");
  }
  hipLaunchKernelGGL(k_fill_u64, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, st, keys, cap, EMPTY_KEY);
  int lg = 0;
  while ((1ll << lg) < cap) ++lg;
  const unsigned int shift = 64u - (unsigned)lg, mask = (unsigned)(cap - 1);
  if (Cn.cap) {
    hipLaunchKernelGGL(k_rehash_corners, dim3((unsigned)((Cn.cap + 255) / 256)), dim3(256), 0, st, Cn.keys, Cn.vals,
                       Cn.cap, keys, vals, shift, mask);
    SHINE_HIP_CHECK(hipGetLastError());
    t->retire(Cn.keys, (size_t)Cn.cap * 8);
    t->retire(Cn.vals, (size_t)Cn.cap * 4);
  }
  Cn.keys = keys;
  Cn.vals = vals;
  Cn.cap = cap;
  Cn.shift = shift;
  Cn.mask = mask;
  return SHINE_OK;
}

static unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

}  // namespace shine

using namespace shine;

extern "C" int shine_tables_insert_corners(shine_tables* t, int32_t slot, const int64_t* corner_keys,
                                           const int32_t* ids, int64_t n, void* stream) {
  if (!t || slot < 0 || slot >= t->n_levels || n < 0) return set_error(SHINE_E_INVALID, "shine_tables_insert_corners: bad table/slot/n");
  if (n == 0) return SHINE_OK;
  if (!corner_keys || !ids) return set_error(SHINE_E_INVALID, "shine_tables_insert_corners: null keys/ids");
  hipStream_t st = (hipStream_t)stream;
  CornerLevel& Cn = t->cl[slot];
  int rc = corner_reserve(t, Cn, Cn.count + n, st);
  if (rc != SHINE_OK) return rc;
  hipLaunchKernelGGL(k_insert_corners, dim3(blocks_for(n)), dim3(256), 0, st, Cn.keys, Cn.vals, Cn.shift, Cn.mask,
                     (const u64*)corner_keys, (const int*)ids, 0, (long long)n);
  SHINE_HIP_CHECK(hipGetLastError());
  Cn.count += n;
  return SHINE_OK;
}

extern "C" int shine_tables_corner_count(const shine_tables* t, int32_t slot, int64_t* count) {
  if (!t || slot < 0 || slot >= t->n_levels || !count) return set_error(SHINE_E_INVALID, "shine_tables_corner_count: bad argument");
  *count = t->cl[slot].count;
  return SHINE_OK;
}

extern "C" int shine_tables_grow(shine_tables* t, const shine_step_config* cfg, const float* points, int64_t n,
                                 int64_t* fresh_counts, int64_t* added_counts, void* stream) {
  if (!t || !cfg || !fresh_counts || !added_counts || n < 0 || (n > 0 && !points))
    return set_error(SHINE_E_INVALID, "shine_tables_grow: null argument");
  const int L = t->n_levels;
  if (cfg->n_levels != L || cfg->max_level < L || cfg->max_level > 15)
    return set_error(SHINE_E_INVALID, "shine_tables_grow: n_levels/max_level mismatch");
  hipStream_t st = (hipStream_t)stream;
  GrowScratch& G = t->grow;
  for (int s = 0; s < L; ++s) {
    fresh_counts[s] = added_counts[s] = 0;
    G.n_fresh[s] = G.n_added[s] = 0;
    G.fresh_keys[s] = nullptr;
    G.fresh_ids[s] = nullptr;
    G.new_corners[s] = nullptr;
  }
  if (n == 0) return SHINE_OK;
  if (n >= (1ll << 31)) return set_error(SHINE_E_INVALID, "shine_tables_grow: more than 2^31 points in one frame");
  SHINE_HIP_CHECK(prim_warmup(st));

  // ---------------- phase A: fresh nodes per level, Morton order
  const unsigned end_bit = 3u * (unsigned)cfg->max_level;
  size_t sort_bytes = 0, scan_bytes = 0;
  SHINE_HIP_CHECK(prim_sort_keys_u64(nullptr, sort_bytes, nullptr, nullptr, (size_t)n, 0u, end_bit, st));
  SHINE_HIP_CHECK(prim_scan_int(nullptr, scan_bytes, nullptr, nullptr, (size_t)n, st));
  SHINE_HIP_CHECK(prim_scan_int(nullptr, scan_bytes, nullptr, nullptr, (size_t)n * (size_t)L, st));  // (all levels in one scan)
  const size_t kb = galign((size_t)n * 8), ib = galign((size_t)n * 4 * (size_t)L);
  const size_t tmp_a = galign(sort_bytes > scan_bytes ? sort_bytes : scan_bytes);
  const size_t need_a = 2 * kb + 2 * ib + tmp_a + (size_t)L * kb + galign(3 * SHINE_MAX_LEVELS * 8);
  int rc = ensure(t, &G.a, &G.a_bytes, need_a, st);
  if (rc != SHINE_OK) return rc;
  char* a = (char*)G.a;
  u64* k0 = (u64*)a;
  u64* k1 = (u64*)(a + kb);
  int* flags = (int*)(a + 2 * kb);
  int* pos = (int*)(a + 2 * kb + ib);
  void* tmp = a + 2 * kb + 2 * ib;
  char* fresh_base = a + 2 * kb + 2 * ib + tmp_a;
  long long* d_counts = (long long*)(fresh_base + (size_t)L * kb);  // fresh counts [SHINE_MAX_LEVELS] | corner starts [L + 1]
  SHINE_HIP_CHECK(hipMemsetAsync(d_counts, 0, 3 * SHINE_MAX_LEVELS * 8, st));

  const float res = (float)(1u << cfg->max_level);
  hipLaunchKernelGGL(k_leaf_keys, dim3(blocks_for(n)), dim3(256), 0, st, points, (long long)n, res, k0);
  SHINE_HIP_CHECK(hipGetLastError());
  SHINE_HIP_CHECK(prim_sort_keys_u64(tmp, sort_bytes, k0, k1, (size_t)n, 0u, end_bit, st));
  {
    LevelProbes P = {};
    for (int s = 0; s < L; ++s) {
      const int level = cfg->max_level - (L - 1 - s);
      P.sh[s] = 3 * (cfg->max_level - level);
      P.T[s] = ProbeTable{t->lv[s].keys, t->lv[s].shift, t->lv[s].mask};
      G.fresh_keys[s] = (u64*)(fresh_base + (size_t)s * kb);
      P.out[s] = G.fresh_keys[s];
    }
    const dim3 grid(blocks_for(n), (unsigned)L);
    hipLaunchKernelGGL(k_flag_fresh_levels, grid, dim3(256), 0, st, k1, (long long)n, P, flags);
    SHINE_HIP_CHECK(prim_scan_int(tmp, scan_bytes, flags, pos, (size_t)n * (size_t)L, st));
    hipLaunchKernelGGL(k_compact_levels, grid, dim3(256), 0, st, k1, (long long)n, P, flags, pos, d_counts);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  long long h_counts[2 * SHINE_MAX_LEVELS];
  SHINE_HIP_CHECK(hipMemcpyAsync(h_counts, d_counts, SHINE_MAX_LEVELS * 8, hipMemcpyDeviceToHost, st));
  SHINE_HIP_CHECK(hipStreamSynchronize(st));
  long long max_nf = 0, sum_nf = 0;
  for (int s = 0; s < L; ++s) {
    G.n_fresh[s] = h_counts[s];
    fresh_counts[s] = h_counts[s];
    sum_nf += h_counts[s];
    if (h_counts[s] > max_nf) max_nf = h_counts[s];
  }
  if (sum_nf == 0) return SHINE_OK;  // nothing new anywhere (:129-130 for every level)

  // ---------------- phase B: new corners per level, lexicographic order — all levels in one sort / flag / scan / compact
  const size_t mc = (size_t)sum_nf * 8;
  size_t csort_bytes = 0, cscan_bytes = 0;
  SHINE_HIP_CHECK(prim_sort_keys_u64(nullptr, csort_bytes, nullptr, nullptr, mc, 0u, 64u, st));
  SHINE_HIP_CHECK(prim_scan_int(nullptr, cscan_bytes, nullptr, nullptr, mc, st));
  const size_t ckb = galign(mc * 8), cib = galign(mc * 4);
  const size_t tmp_b = galign(csort_bytes > cscan_bytes ? csort_bytes : cscan_bytes);
  size_t need_b = 3 * ckb + 2 * cib + tmp_b;  // tagged keys (x 2, sort), new corners of all levels, flags, positions
  size_t off_ids[SHINE_MAX_LEVELS];
  for (int s = 0; s < L; ++s) {
    off_ids[s] = need_b;
    need_b += galign((size_t)G.n_fresh[s] * 8 * 4);
  }
  rc = ensure(t, &G.b, &G.b_bytes, need_b, st);
  if (rc != SHINE_OK) return rc;
  char* b = (char*)G.b;
  u64* c0 = (u64*)b;
  u64* c1 = (u64*)(b + ckb);
  u64* new_all = (u64*)(b + 2 * ckb);
  int* cflags = (int*)(b + 3 * ckb);
  int* cpos = (int*)(b + 3 * ckb + cib);
  void* ctmp = b + 3 * ckb + 2 * cib;
  long long* d_seg = d_counts + SHINE_MAX_LEVELS;  // [L + 1] starts of the levels' new corners (SHINE_MAX_LEVELS + 1 <= 16)
  {
    ExpandLevels E = {};
    CornerProbes CP = {};
    long long off = 0;
    for (int s = 0; s < L; ++s) {
      E.fresh[s] = G.fresh_keys[s];
      E.nf[s] = G.n_fresh[s];
      E.off[s] = off;
      off += G.n_fresh[s] * 8;
      CP.T[s] = ProbeTable{t->cl[s].keys, t->cl[s].shift, t->cl[s].mask};
      G.fresh_ids[s] = G.n_fresh[s] ? (int*)(b + off_ids[s]) : nullptr;
    }
    const long long m = (long long)mc;
    hipLaunchKernelGGL(k_expand_corners_levels, dim3(blocks_for(max_nf * 8), (unsigned)L), dim3(256), 0, st, E, c0);
    size_t sb = csort_bytes, cb = cscan_bytes;
    SHINE_HIP_CHECK(prim_sort_keys_u64(ctmp, sb, c0, c1, mc, 0u, 64u, st));
    hipLaunchKernelGGL(k_flag_fresh_corners, dim3(blocks_for(m)), dim3(256), 0, st, c1, m, CP, cflags);
    SHINE_HIP_CHECK(prim_scan_int(ctmp, cb, cflags, cpos, mc, st));
    hipLaunchKernelGGL(k_compact_corners, dim3(blocks_for(m)), dim3(256), 0, st, c1, m, L, cflags, cpos, new_all, d_seg);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  long long h_seg[SHINE_MAX_LEVELS + 1];
  SHINE_HIP_CHECK(hipMemcpyAsync(h_seg, d_seg, (size_t)(L + 1) * 8, hipMemcpyDeviceToHost, st));
  SHINE_HIP_CHECK(hipStreamSynchronize(st));
  for (int s = 0; s < L; ++s) {
    h_counts[SHINE_MAX_LEVELS + s] = h_seg[s + 1] - h_seg[s];
    G.new_corners[s] = G.n_fresh[s] ? new_all + h_seg[s] : nullptr;
  }

  // ---------------- phase C: inserts
  for (int s = 0; s < L; ++s) {
    const long long nf = G.n_fresh[s];
    if (nf == 0) continue;
    const long long added = h_counts[SHINE_MAX_LEVELS + s];
    G.n_added[s] = added;
    added_counts[s] = added;
    CornerLevel& Cn = t->cl[s];
    if (Cn.count + added >= (1ll << 29)) return set_error(SHINE_E_INVALID, "shine_tables_grow: level exceeds 2^29 rows");
    rc = corner_reserve(t, Cn, Cn.count + added, st);
    if (rc != SHINE_OK) return rc;
    if (added > 0) {
      hipLaunchKernelGGL(k_insert_corners, dim3(blocks_for(added)), dim3(256), 0, st, Cn.keys, Cn.vals, Cn.shift,
                         Cn.mask, G.new_corners[s], (const int*)nullptr, (int)Cn.count, added);
      SHINE_HIP_CHECK(hipGetLastError());
      Cn.count += added;
    }
    ProbeTable T = {Cn.keys, Cn.shift, Cn.mask};
    hipLaunchKernelGGL(k_corner_ids, dim3(blocks_for(nf * 8)), dim3(256), 0, st, G.fresh_keys[s], nf, T, Cn.vals,
                       G.fresh_ids[s]);
    SHINE_HIP_CHECK(hipGetLastError());
    rc = shine_tables_insert(t, s, (const int64_t*)G.fresh_keys[s], G.fresh_ids[s], nf, stream);
    if (rc != SHINE_OK) return rc;
  }
  return SHINE_OK;
}

extern "C" int shine_tables_grow_fetch(const shine_tables* t, int32_t slot, int64_t* fresh_keys, int32_t* fresh_ids,
                                       int64_t* new_corner_keys, void* stream) {
  if (!t || slot < 0 || slot >= t->n_levels) return set_error(SHINE_E_INVALID, "shine_tables_grow_fetch: bad table/slot");
  const GrowScratch& G = t->grow;
  hipStream_t st = (hipStream_t)stream;
  const long long nf = G.n_fresh[slot], na = G.n_added[slot];
  if (fresh_keys && nf) SHINE_HIP_CHECK(hipMemcpyAsync(fresh_keys, G.fresh_keys[slot], (size_t)nf * 8, hipMemcpyDeviceToDevice, st));
  if (fresh_ids && nf) SHINE_HIP_CHECK(hipMemcpyAsync(fresh_ids, G.fresh_ids[slot], (size_t)nf * 32, hipMemcpyDeviceToDevice, st));
  if (new_corner_keys && na)
    SHINE_HIP_CHECK(hipMemcpyAsync(new_corner_keys, G.new_corners[slot], (size_t)na * 8, hipMemcpyDeviceToDevice, st));
  return SHINE_OK;
}

// everything the last shine_tables_grow added, all levels, in ONE launch: per level {keys [fresh] | ids [fresh][8] as 4 words
// per node | new corner keys [added]} as 8-byte words, level after level
struct FetchAll {
  const unsigned long long* src[3 * SHINE_MAX_LEVELS];
  long long begin[3 * SHINE_MAX_LEVELS + 1];  // word offsets into the output
  int segs;
};
__global__ __launch_bounds__(256) void k_fetch_all(FetchAll f, unsigned long long* __restrict__ out) {
  const long long total = f.begin[f.segs];
  for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < total; w += (long long)gridDim.x * 256) {
    int k = 0;
    while (w >= f.begin[k + 1]) ++k;
    out[w] = f.src[k][w - f.begin[k]];
  }
}

extern "C" int shine_tables_grow_fetch_all(const shine_tables* t, int64_t* out, int64_t out_words, void* stream) {
  if (!t) return set_error(SHINE_E_INVALID, "shine_tables_grow_fetch_all: null tables");
  const GrowScratch& G = t->grow;
  FetchAll f = {};
  long long w = 0;
  for (int s = 0; s < t->n_levels; ++s) {
    const long long nf = G.n_fresh[s], na = G.n_added[s];
    const unsigned long long* src[3] = {G.fresh_keys[s], reinterpret_cast<const unsigned long long*>(G.fresh_ids[s]),
                                        G.new_corners[s]};
    const long long len[3] = {nf, 4 * nf, na};
    for (int k = 0; k < 3; ++k) {
      f.src[f.segs] = src[k];
      f.begin[f.segs] = w;
      w += len[k];
      ++f.segs;
    }
  }
  f.begin[f.segs] = w;
  if (w == 0) return SHINE_OK;
  if (!out || out_words < w) return set_error(SHINE_E_INVALID, "shine_tables_grow_fetch_all: output too small");
  long long blocks = (w + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(k_fetch_all, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, f,
                     reinterpret_cast<unsigned long long*>(out));
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_tables_rank_nodes(shine_tables* t, int64_t* n_buckets_out, void* stream) {
  if (!t) return set_error(SHINE_E_INVALID, "shine_tables_rank_nodes: null tables");
  const int L = t->n_levels;
  hipStream_t st = (hipStream_t)stream;
  long long total_cap = 0, total_nodes = 0;
  for (int s = 0; s < L; ++s) {
    if (!t->lv[s].keys) return set_error(SHINE_E_STATE, "shine_tables_rank_nodes: a featured level has no table yet");
    total_cap += t->lv[s].cap;
    total_nodes += t->lv[s].count;
  }
  if (total_cap >= (1ll << 31)) return set_error(SHINE_E_INVALID, "shine_tables_rank_nodes: more than 2^31 slots");
  size_t sort_bytes = 0;
  SHINE_HIP_CHECK(prim_sort_pairs_u64(nullptr, sort_bytes, nullptr, nullptr, nullptr, nullptr, (size_t)total_cap, 0u, 64u, st));
  const size_t kb = galign((size_t)total_cap * 8);
  GrowScratch& G = t->grow;
  int rc = ensure(t, &G.b, &G.b_bytes, 4 * kb + galign(sort_bytes), st);  // grow results in b are consumed by now
  if (rc != SHINE_OK) return rc;
  for (int s = 0; s < SHINE_MAX_LEVELS; ++s) {  // ...and no longer fetchable
    G.n_fresh[s] = G.n_added[s] = 0;
  }
  char* b = (char*)G.b;
  u64 *rk0 = (u64*)b, *rk1 = (u64*)(b + kb), *rv0 = (u64*)(b + 2 * kb), *rv1 = (u64*)(b + 3 * kb);
  long long off = 0;
  RankPtrs P = {};
  for (int s = 0; s < L; ++s) {
    const TableLevel& T = t->lv[s];
    const int sh = 3 * (L - 1 - s);
    hipLaunchKernelGGL(k_rank_keys, dim3(blocks_for(T.cap)), dim3(256), 0, st, T.keys, (long long)T.cap, sh, L - 1 - s, s,
                       rk0 + off, rv0 + off);
    P.ranks[s] = T.ranks;
    off += T.cap;
  }
  SHINE_HIP_CHECK(hipGetLastError());
  SHINE_HIP_CHECK(prim_sort_pairs_u64(b + 4 * kb, sort_bytes, rk0, rk1, rv0, rv1, (size_t)total_cap, 0u, 64u, st));
  if (total_nodes)
    hipLaunchKernelGGL(k_rank_scatter, dim3(blocks_for(total_nodes)), dim3(256), 0, st, rv1, total_nodes, P);
  SHINE_HIP_CHECK(hipGetLastError());
  t->n_buckets = total_nodes + 64;  // + shine_plan.hip MISS_BUCKETS
  if (n_buckets_out) *n_buckets_out = t->n_buckets;
  return SHINE_OK;
}
