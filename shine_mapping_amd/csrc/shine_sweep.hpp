// shine_sweep.hpp — argument blocks of the batched importance sweep (shine_sweep.hip)
#pragma once
#include "shine_step_common.hpp"

namespace shine {

constexpr int SWEEP_GROUP = 64;  // chunks of one launch (their private gradient tables live side by side in the scratch)

constexpr long long SWEEP_SORT_MAX = 16384;  // largest chunk (bs) one workgroup sorts in LDS (64 KB of positions)

struct SweepArgs {
  V1Args a;                         // what all chunks share; a.lv[].grad / a.touched[] = the FIRST chunk's tables / flags
  long long begin[SWEEP_GROUP + 1];  // the group's chunk boundaries in a.perm
  long long grad_stride;            // floats from one chunk's tables to the next chunk's
  long long flag_stride;            // bytes, likewise
  int bpc;                          // workgroups per chunk
};

struct SweepFoldArgs {
  float4* imp[LCAP];
  float4* feat[LCAP];
  float* grad[LCAP];
  unsigned char* flags[LCAP];
  long long rows[LCAP];
  long long grad_stride, flag_stride;
  int chunks;
};

}  // namespace shine
