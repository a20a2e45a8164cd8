// shine_exchange.hip — device side of the data-parallel "touched rows" gradient exchange (exchange only the
// feature rows a step touched instead of the dense tables).  The reference is single-GPU (utils/tools.py:26): no
// counterpart there.  Per step and rank:
//   shine_touched_index   byte flags (shine_mark_touched, OR-reduced over the ranks) -> ascending row-id lists per level
//                         (exclusive scan + compaction: the same order on every rank) and their counts
//   shine_touched_pack    the listed rows of every level's dense gradient, then the L trash rows -> one message
//   [RCCL all-reduce of the message, torch.distributed]
//   shine_touched_unpack  the reverse, and the flags are cleared for the next step
// and the collective-light form of the same exchange (shine_rows_*, below): every rank packs only the rows IT touched (ids +
// values, fixed capacity, counts stay on the device) and the ranks all-gather their messages — one collective, no host read,
// capturable in a HIP graph, half the bytes of an all-reduce of the union.
#include <cstring>
#include "shine_internal.hpp"

namespace shine {

__global__ void k_touched_compact(const unsigned char* flags, const int* pos, long long n, int* idx_out,
                                  long long* count_out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = flags[i] ? 1 : 0;
  if (f) idx_out[pos[i]] = (int)i;
  if (i == n - 1) *count_out = (long long)pos[i] + f;
}

struct ExchangeArgs {
  float* grad[SHINE_MAX_LEVELS];        // dense gradient tables [rows + 1][F]
  const int* idx[SHINE_MAX_LEVELS];     // ascending touched row ids
  unsigned char* flags[SHINE_MAX_LEVELS];
  long long off[SHINE_MAX_LEVELS + 1];  // first message row of each level; off[L] = total rows
  long long rows[SHINE_MAX_LEVELS];     // trash row index of each level
  int n_levels;
};

// one thread per (message row, feature): rows of level l occupy message rows [off[l], off[l+1]); the L trash rows follow
template <bool PACK>
__global__ void k_touched_move(ExchangeArgs a, float* msg) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long r = t >> 3;
  const int q = (int)(t & 7);
  const long long total = a.off[a.n_levels];
  if (r >= total + a.n_levels) return;
  float* g;
  if (r >= total) {  // trash rows: every miss of the step lands there
    const int l = (int)(r - total);
    g = a.grad[l] + a.rows[l] * F + q;
  } else {
    int l = 0;
#pragma unroll
    for (int k = 1; k < SHINE_MAX_LEVELS; ++k)
      if (k < a.n_levels && r >= a.off[k]) l = k;
    const int row = a.idx[l][r - a.off[l]];
    g = a.grad[l] + (long long)row * F + q;
    if (!PACK && q == 0 && a.flags[l]) a.flags[l][row] = 0;  // cleared for the next step
  }
  if (PACK) msg[t] = *g;
  else *g = msg[t];
}

static size_t xalign(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace shine

using namespace shine;

extern "C" int shine_touched_index(int32_t n_levels, const uint8_t* const* flags, const int64_t* rows,
                                   int32_t* const* idx_out, int64_t* counts_dev, void* workspace, size_t* workspace_bytes,
                                   void* stream) {
  if (n_levels < 1 || n_levels > SHINE_MAX_LEVELS || !rows || !workspace_bytes)
    return set_error(SHINE_E_INVALID, "shine_touched_index: bad argument");
  hipStream_t st = (hipStream_t)stream;
  long long max_rows = 0;
  for (int l = 0; l < n_levels; ++l) max_rows = rows[l] > max_rows ? rows[l] : max_rows;
  size_t scan_bytes = 0;
  SHINE_HIP_CHECK(prim_scan_flags(nullptr, scan_bytes, nullptr, nullptr, (size_t)(max_rows > 0 ? max_rows : 1), st));
  const size_t need = xalign(scan_bytes) + xalign((size_t)max_rows * sizeof(int));
  if (!workspace) {
    *workspace_bytes = need;
    return SHINE_OK;
  }
  if (*workspace_bytes < need) return set_error(SHINE_E_INVALID, "shine_touched_index: workspace too small");
  if (!flags || !idx_out || !counts_dev) return set_error(SHINE_E_INVALID, "shine_touched_index: null argument");
  char* tmp = (char*)workspace;
  int* pos = (int*)(tmp + xalign(scan_bytes));
  for (int l = 0; l < n_levels; ++l) {
    const long long n = rows[l];
    if (n <= 0) {
      SHINE_HIP_CHECK(hipMemsetAsync(counts_dev + l, 0, sizeof(int64_t), st));
      continue;
    }
    size_t sb = scan_bytes;
    SHINE_HIP_CHECK(prim_scan_flags(tmp, sb, (const unsigned char*)flags[l], pos, (size_t)n, st));
    hipLaunchKernelGGL(k_touched_compact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const unsigned char*)flags[l], (const int*)pos, n, (int*)idx_out[l], (long long*)(counts_dev + l));
    SHINE_HIP_CHECK(hipGetLastError());
  }
  return SHINE_OK;
}

static int touched_move(bool pack, int32_t n_levels, float* const* grads, const int32_t* const* idx,
                        const int64_t* counts, const int64_t* rows, uint8_t* const* flags, float* msg, void* stream) {
  if (n_levels < 1 || n_levels > SHINE_MAX_LEVELS || !grads || !idx || !counts || !rows || !msg)
    return set_error(SHINE_E_INVALID, "shine_touched_pack/unpack: bad argument");
  ExchangeArgs a = {};
  a.n_levels = n_levels;
  long long off = 0;
  for (int l = 0; l < n_levels; ++l) {
    a.grad[l] = grads[l];
    a.idx[l] = idx[l];
    a.flags[l] = flags ? flags[l] : nullptr;
    a.rows[l] = rows[l];
    a.off[l] = off;
    off += counts[l];
  }
  a.off[n_levels] = off;
  const long long threads = (off + n_levels) * F;
  const dim3 grid((unsigned)((threads + 255) / 256));
  if (pack) hipLaunchKernelGGL(k_touched_move<true>, grid, dim3(256), 0, (hipStream_t)stream, a, msg);
  else hipLaunchKernelGGL(k_touched_move<false>, grid, dim3(256), 0, (hipStream_t)stream, a, msg);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_touched_pack(int32_t n_levels, float* const* grads, const int32_t* const* idx, const int64_t* counts,
                                  const int64_t* rows, float* msg, void* stream) {
  return touched_move(true, n_levels, grads, idx, counts, rows, nullptr, msg, stream);
}

extern "C" int shine_touched_unpack(int32_t n_levels, float* const* grads, const int32_t* const* idx, const int64_t* counts,
                                    const int64_t* rows, uint8_t* const* flags, const float* msg, void* stream) {
  return touched_move(false, n_levels, grads, idx, counts, rows, flags, const_cast<float*>(msg), stream);
}


// ------------------------------------------------------------------------------------------------------------------------
// Own-rows all-gather exchange.  The gradient bucket is ONE flat array (the L dense feature-grad tables [rows_l + 1][8] back
// to back, then a dense tail: the decoder's 1377 floats) and `flags` ONE byte per 8-float row of its table part (the step's
// touched flags of all levels, back to back).  Under the sorted global draw a rank's slice is a contiguous stretch of the
// node-ordered batch, so it touches ~1/world of the rows: instead of all-reducing the union of all ranks' rows, every rank
//   shine_rows_pack        compacts ITS flagged rows (ascending) and MOVES them out of the bucket into a fixed-size message
//                          msg = {count, overflow, 0, 0 | ids[cap] | values[cap][8] | tail[tail_n]}  (4-byte words),
//                          zeroing the moved rows (and the tail) and clearing the flags (so a second micro-batch of the same
//                          step can accumulate into the same bucket and be packed on its own, without double counting);
//   [all-gather of the messages: torch.distributed / RCCL]
//   shine_rows_unpack_add  adds every rank's rows (its own included) back into the bucket, rank after rank, and sums the
//                          dense tails in rank order: every rank ends up with bit-identical sums.
// A message holds at most `cap` rows; a rank that touched more sets msg[1] = 1 and every rank sees it (overflow_out).
namespace shine {

constexpr int ROWS_HDR = 4;  // message header words: count, overflow, reserved x 2

// one thread per row: the flagged rows move into the message (ids ascending = scan order)
struct KeepRows {
  long long row[SHINE_MAX_LEVELS];  // rows whose flag stays set (the trash rows: every miss of every step lands there), or -1
};

__global__ void k_rows_pack(unsigned char* flags, const int* pos, long long n_rows, float* bucket, int* msg, long long cap,
                            KeepRows keep) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const int f = flags[i] ? 1 : 0;
  const long long k = pos[i];
  if (i == n_rows - 1) {
    const long long total = k + f;
    msg[0] = (int)(total < cap ? total : cap);
    msg[1] = total > cap ? 1 : 0;
    msg[2] = msg[3] = 0;
  }
  if (!f) return;
  bool sticky = false;
#pragma unroll
  for (int l = 0; l < SHINE_MAX_LEVELS; ++l) sticky |= keep.row[l] == i;
  if (!sticky) flags[i] = 0;  // cleared for the next step
  if (k >= cap) return;  // overflow: reported through msg[1], the row stays in the bucket
  msg[ROWS_HDR + k] = (int)i;
  float4* src = reinterpret_cast<float4*>(bucket + i * F);
  float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(msg) + ROWS_HDR + cap + k * F);
  dst[0] = src[0];
  dst[1] = src[1];
  src[0] = src[1] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void k_rows_pack_tail(float* bucket_tail, float* msg_tail, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {  // moved like the rows: a later micro-batch of the step accumulates from zero
    msg_tail[i] = bucket_tail[i];
    bucket_tail[i] = 0.f;
  }
}

// ONE rank's message per launch, the launches in rank order: the ids of a message are distinct, so the adds need no atomics,
// and every rank applies the same contributions in the same order — the replicas stay bit-identical (fp32 atomics over all
// ranks at once would be one launch, but rows shared by three or more ranks — the coarse levels — would then sum in a
// different order on every rank).  One thread per (row slot, feature).
__global__ void k_rows_unpack_add(const int* m, long long cap, float* bucket, int* overflow_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long k = t >> 3;
  const int q = (int)(t & 7);
  if (t == 0 && m[1] && overflow_out) *overflow_out = 1;
  if (k >= m[0]) return;
  const long long row = m[ROWS_HDR + k];
  bucket[row * F + q] += reinterpret_cast<const float*>(m)[ROWS_HDR + cap + k * F + q];
}

__global__ void k_rows_unpack_tail(const int* msgs, long long msg_words, long long tail_off_words, int world,
                                   float* bucket_tail, long long n, int add) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int r = 0; r < world; ++r) s += reinterpret_cast<const float*>(msgs + (long long)r * msg_words)[tail_off_words + i];
  bucket_tail[i] = add ? bucket_tail[i] + s : s;  // (the ranks' sum first, then the bucket: the same order on every rank)
}

}  // namespace shine

extern "C" int64_t shine_rows_message_words(int64_t cap, int64_t tail_n) {
  return cap < 0 || tail_n < 0 ? -1 : ROWS_HDR + cap + cap * F + tail_n;
}

extern "C" int shine_rows_pack(uint8_t* flags, int64_t n_rows, const int64_t* keep_rows, int32_t n_keep, float* bucket,
                               int64_t tail_off, int64_t tail_n, int64_t cap, int32_t* msg, void* workspace,
                               size_t* workspace_bytes, void* stream) {
  if (n_keep < 0 || n_keep > SHINE_MAX_LEVELS || (n_keep > 0 && !keep_rows))
    return set_error(SHINE_E_INVALID, "shine_rows_pack: bad keep_rows");
  KeepRows keep;
  for (int l = 0; l < SHINE_MAX_LEVELS; ++l) keep.row[l] = l < n_keep ? keep_rows[l] : -1;
  if (n_rows < 1 || n_rows > 0x7fffffffll || cap < 1 || tail_n < 0 || !workspace_bytes)
    return set_error(SHINE_E_INVALID, "shine_rows_pack: bad argument");
  hipStream_t st = (hipStream_t)stream;
  size_t scan_bytes = 0;
  SHINE_HIP_CHECK(prim_scan_flags(nullptr, scan_bytes, nullptr, nullptr, (size_t)n_rows, st));
  const size_t need = xalign(scan_bytes) + xalign((size_t)n_rows * sizeof(int));
  if (!workspace) {
    *workspace_bytes = need;
    return SHINE_OK;
  }
  if (*workspace_bytes < need) return set_error(SHINE_E_INVALID, "shine_rows_pack: workspace too small");
  if (!flags || !bucket || !msg) return set_error(SHINE_E_INVALID, "shine_rows_pack: null argument");
  if (((size_t)bucket | (size_t)msg) & 15) return set_error(SHINE_E_INVALID, "shine_rows_pack: bucket and msg must be 16-byte aligned");
  if (cap & 3) return set_error(SHINE_E_INVALID, "shine_rows_pack: cap must be a multiple of 4 (16-byte aligned value rows)");
  char* tmp = (char*)workspace;
  int* pos = (int*)(tmp + xalign(scan_bytes));
  size_t sb = scan_bytes;
  SHINE_HIP_CHECK(prim_scan_flags(tmp, sb, (const unsigned char*)flags, pos, (size_t)n_rows, st));
  hipLaunchKernelGGL(k_rows_pack, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, st, (unsigned char*)flags,
                     (const int*)pos, (long long)n_rows, bucket, (int*)msg, (long long)cap, keep);
  SHINE_HIP_CHECK(hipGetLastError());
  if (tail_n > 0) {
    hipLaunchKernelGGL(k_rows_pack_tail, dim3((unsigned)((tail_n + 255) / 256)), dim3(256), 0, st, bucket + tail_off,
                       reinterpret_cast<float*>(msg) + ROWS_HDR + cap + cap * F, (long long)tail_n);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  return SHINE_OK;
}

extern "C" int shine_rows_unpack_add(const int32_t* msgs, int32_t world, int64_t cap, float* bucket, int64_t tail_off,
                                     int64_t tail_n, int32_t tail_add, int32_t* overflow_out, void* stream) {
  if (!msgs || world < 1 || cap < 1 || !bucket || tail_n < 0)
    return set_error(SHINE_E_INVALID, "shine_rows_unpack_add: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long long words = ROWS_HDR + cap + cap * F + tail_n;
  for (int r = 0; r < world; ++r) {
    hipLaunchKernelGGL(k_rows_unpack_add, dim3((unsigned)((cap * F + 255) / 256)), dim3(256), 0, st,
                       (const int*)msgs + (long long)r * words, (long long)cap, bucket, (int*)overflow_out);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  if (tail_n > 0) {
    hipLaunchKernelGGL(k_rows_unpack_tail, dim3((unsigned)((tail_n + 255) / 256)), dim3(256), 0, st, (const int*)msgs, words,
                       (long long)(ROWS_HDR + cap + cap * F), (int)world, bucket + tail_off, (long long)tail_n, (int)(tail_add != 0));
    SHINE_HIP_CHECK(hipGetLastError());
  }
  return SHINE_OK;
}
