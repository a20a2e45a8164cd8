// shine_finish_args.hpp — the argument block of the iteration tail (shine_finish.hip k_finish) and its prepared launch, shared
// with the iteration graph (shine_graph.hip), which makes the launch a kernel node.
#pragma once
#include "shine_step_common.hpp"

namespace shine {

constexpr int FIN_MAX_SEG = 16;

struct FinSeg {
  float* p;
  float* g;
  float* m;
  float* v;
  long long n;       // elements
  long long ustart;  // first work unit (8 floats) of the segment
  float wd;
  int lr_idx;        // this tensor's entry of lr_dev
  int part_off;      // decoder tensor: index of its element 0 in the partial vector; feature table: -1
};

struct FinArgs {
  FinSeg seg[FIN_MAX_SEG];
  int n_seg, n_levels;
  long long feat_units;  // units of the feature tables (= rows incl. the trash rows); the decoder's units follow
  long long dec_units;
  // the pending step
  const float* partials;
  int nblocks;
  const long long* n_surf;
  int n_surf_parts;
  int reduction_sum;
  float inv_n, weight_e;
  double* loss_parts;
  // regulariser (lambda == 0: off)
  const float* last[SHINE_MAX_LEVELS];
  const float* imp[SHINE_MAX_LEVELS];
  unsigned char* touched[SHINE_MAX_LEVELS];
  int grad_on[SHINE_MAX_LEVELS];
  float lambda;
  double* reg_out;
  // active rows (exact): touched[s][r] == 0 means "no gradient since the optimiser was created" — then m = v = g = 0 and torch's
  // Adam (no weight decay on the feature tables) computes p -= lr * 0 / (0 + eps): the row is left bit for bit as it is, so it is
  // not even read.  The step's scatter sets 1 on the rows of THIS iteration (what the regulariser applies to), this launch turns
  // 1 into 2 ("touched earlier"): the flags are sticky until the caller clears them together with the optimiser state.
  int active;
  // Adam
  float b1, b2, eps;
  const long long* step_state;  // already advanced for this step (by the fused kernel, cfg->adam_state)
  const float* lr_dev;
  // the sorted draw of the NEXT iteration (shine_next_draw), by nd_blocks extra blocks; nd_blocks == 0: none
  int nd_blocks;
  long long nd_n, nd_pool;
  unsigned long long nd_seed;
  unsigned long long* nd_stream;
  int* nd_idx;
  const unsigned int* nd_bits;
  long long* nd_surf;
  // the decoder's MFMA operand image (shine_tile16.hpp: V3_IMAGE_FLOATS floats) or null: every decoder element this launch
  // updates is also written to its place(s) in the image, which the next small-batch step copies instead of rebuilding it
  float* op_image;
};

// a tail launch, prepared but not launched (shine_finish.hip prepare_finish)
struct FinLaunch {
  const void* fn;
  dim3 grid, block;
  FinArgs a;
  int fb, db;
};
int prepare_finish(FinLaunch* out, const shine_step_config* cfg, int64_t n, const void* workspace, const int64_t* n_surf,
                   double* loss_parts, const float* const* feats_last, const float* const* importance,
                   unsigned char* const* touched, const int32_t* grad_on, float lambda_forget, double* reg_out,
                   int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const int64_t* numel, const float* lr_dev, const int32_t* lr_index,
                   const float* weight_decay, float beta1, float beta2, float eps, const int64_t* step_state,
                   const shine_next_draw* next_draw, int32_t active_rows);

}  // namespace shine
