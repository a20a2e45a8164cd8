// shine_step_common.hpp — what the fused-step kernel (shine_step_v3.hip) and its support kernels
// (shine_step_support.hip) share: the kernel argument block, the layout of the per-workgroup partial vector that
// k_reduce_partials adds up, and a few wave-level device helpers.
#pragma once
#include "shine_internal.hpp"

namespace shine {

constexpr int LCAP = 4;        // featured levels handled by the MFMA kernels (tree_level_feat <= 4 in every yaml)

constexpr int PART_TRASH = SHINE_MLP_PARAMS;                     // + s*8 + q
constexpr int PART_FLOATS = PART_TRASH + SHINE_MAX_LEVELS * 8;   // 1441
constexpr int PART_LOSS = 1444;   // float index of double[3] {bce sum, count, eikonal sum} (8-B aligned)
constexpr int PART_STRIDE = 1456;

// what the hot loop needs per level, nothing else (SGPR budget)
struct V1Level {
  const unsigned long long* keys;
  const int4* vals;
  const float* feat;
  float* grad;
  unsigned int shift, mask;
  float res;
  int pad;
};

struct V1Args {
  V1Level lv[LCAP];
  long long rows[LCAP];
  float* feat_rw[LCAP];  // same tables, writable: the trash row is re-zeroed in-kernel (set_zero, :78-81)
  unsigned char* touched[LCAP];  // optional byte flag per row that received gradient (for shine_regularize)
  const float* coord;
  const float* label;
  const float* weight;
  const int* perm;
  const int* slots;  // [n][L] hash slots per point IN VISITING ORDER (shine_plan_batch), or null: probe in-kernel
  const long long* n_surf;  // [n_surf_parts] partial counts of the batch's surface samples (eikonal), summed by the kernels
  int n_surf_parts;
  int defer_reduce;         // cfg->defer_reduce: no reduction launch; the fused kernel serves the iteration hooks itself
  const float* ext_delta;  // Tier A backward (shine_interp_sdf_backward): d loss / d pred per point, indexed like pred;
                           // the kernel then skips its own loss and backpropagates this instead
  const float* ext_q;      // ... and d loss / d g [n, 3] for g = sigma d pred / d coord (the eikonal build of that backward)
  const float* mlp[6];
  float* pred;
  float* grad_x;
  float* grad_mlp[6];
  double* loss_parts;
  float* partials;
  long long* adam_state;  // iteration hooks (cfg->adam_state / zero_f64): housekeeping of the calls that follow the step,
  double* zero_f64;       // done by one thread of the reduction launch
  float adam_b1, adam_b2;
  long long* prof;  // debug: per-wave phase cycle counters [waves][8] (shine_debug_set_profile_buffer) or null
  long long n;
  // The decoder's MFMA operand image in LDS order (V3_OPTOTAL + 100 floats: what decoder_operands_issue / _store build per
  // workgroup), kept current by the iteration tail, or null: built in-kernel.  Read by the 4-wave (small-batch) builds only; set by
  // the iteration graph only (shine_graph.hip), which owns the buffer.  (In the place of a retired 8-byte field: the argument
  // block keeps its size, so the hidden arguments behind it — and with them every kernel's code — stay where they were.)
  const float* op_image;
  long long tiles;        // tiles of the launch (16 points each) ...
  long long waves_total;  // ... dealt evenly to this many waves: wave w owns tiles [w T / W, (w + 1) T / W)
  int n_levels;
  int reduction_sum;
  int decoder_grad_on;
  int poly;
  int weighted;   // 1: every sample's BCE term (and its gradient) is multiplied by |weight| (utils/loss.py:18-19)
  int pool_mode;  // 1: coord/label/weight/slots are a node-ordered POOL indexed by perm[i] (sorted sample indices,
                  //    shine_sample_sorted); pred / grad_x are written at the batch position i.  0: a batch.
                  // 2: the same with the pool as ONE 32-byte record per sample (cfg->sorted_input 3, shine_step_body.hpp
                  //    RecLayout): coord = the record base, label / slots unused, weight a separate array for L = 4 only
  int ablate;  // debug only (kernel_variant >> 8): 1 no feature atomics, 2 no weight-grad phase, 4 no scatter phase,
               // 8 no row gathers, 16 no probe (every point misses), 32 skip the partial-sum reduction launch,
               // 64 deterministic accumulation: one wave of one workgroup walks the whole stream (tests)
  float sigma;
  float inv_n;
  float weight_e;
};

// SLICED builds of the step body (shine_sweep.hip: many independent small steps in ONE launch): what differs from one slice of
// the launch to the next, everything else comes from the shared V1Args
struct StepSlice {
  const int* perm;   // the slice's sample indices
  long long n;       // ... and their number
  float inv_n;       // the slice's own normaliser
  int part_bid;      // index of the workgroup's partial vector in the workspace
  float* grad[LCAP];             // the slice's private gradient tables
  unsigned char* touched[LCAP];  // ... and touched-row flags (MARK builds)
};

__device__ __forceinline__ long long clk() { return (long long)__builtin_readcyclecounter(); }

// sum over the 16 lanes of a DPP row, result in every lane of the row (4 VALU ops, no LDS traffic)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// d w_c / d x for ONE corner (keeps the eikonal path's live set small; corner_weight_grads builds all 24 values)
__device__ __forceinline__ void corner_dw(const Axis& X, const Axis& Y, const Axis& Z, int c, float out[3]) {
  const int cx = (c >> 2) & 1, cy = (c >> 1) & 1, cz = c & 1;
  const float px = cx ? X.t : 1.0f - X.t, py = cy ? Y.t : 1.0f - Y.t, pz = cz ? Z.t : 1.0f - Z.t;
  const float gx = cx ? X.dt : -X.dt, gy = cy ? Y.dt : -Y.dt, gz = cz ? Z.dt : -Z.dt;
  out[0] = gx * py * pz;
  out[1] = px * gy * pz;
  out[2] = px * py * gz;
}

// ... and for the corner PAIR (2 cp, 2 cp + 1) — the same (cx, cy), cz = 0 / 1 — as packed values: out[e] = {d w_{2cp} / d x_e,
// d w_{2cp+1} / d x_e}, every product in corner_dw's association
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void corner_dw_pair(const Axis& X, const Axis& Y, const Axis& Z, int cp, f32x2_t out[3]) {
  const int cx = (cp >> 1) & 1, cy = cp & 1;
  const float px = cx ? X.t : 1.0f - X.t, py = cy ? Y.t : 1.0f - Y.t;
  const float gx = cx ? X.dt : -X.dt, gy = cy ? Y.dt : -Y.dt;
  const f32x2_t pz = {1.0f - Z.t, Z.t}, gz = {-Z.dt, Z.dt};
  const float a = gx * py, b = px * gy, c = px * py;
  out[0] = (f32x2_t){a, a} * pz;
  out[1] = (f32x2_t){b, b} * pz;
  out[2] = (f32x2_t){c, c} * gz;
}

__device__ __forceinline__ Axis axis_weight_rt(bool poly, float x, float res) {
  return poly ? axis_weight<true>(x, res, res * 0.5f) : axis_weight<false>(x, res, res * 0.5f);
}

// launch geometry of a 16-point-tile kernel
struct V2Geometry {
  long long waves, chunk, blocks, tiles;
  int wg_waves;  // waves per workgroup
};
V2Geometry v3_geometry(long long n, int ablate = 0);  // shine_step_v3.hip: tiles dealt evenly to every resident wave slot
long long v3_lds_bytes(int wg_waves);

// measurement aid (shine_debug_set_profile_buffer): per-wave phase cycle counters or null
extern long long* g_prof_buffer;

// a fused-step launch, prepared but not launched (shine_step_v3.hip prepare_step_v3): shine_train_step_v3 launches it, the
// iteration graph (shine_graph.hip) makes it a kernel node
struct StepLaunch {
  const void* fn;   // the k_step_v3 instantiation; null: empty batch, nothing to launch
  dim3 grid, block;
  int blocks;       // workgroups = partial vectors the launch leaves in the workspace
  bool mark_pass;   // the touched-row flags need k_mark_touched in front (a level without a gradient table, profiling build)
  V1Args a;
};
int prepare_step_v3(StepLaunch* out, const shine_tables* t, const shine_step_config* cfg, const float* coord,
                    const float* sdf_label, const float* weight, const int32_t* perm, const int32_t* slots,
                    const int64_t* n_surf, int64_t n, const float* const* feats, const int64_t* rows, const float* const* mlp,
                    float* pred_out, float* grad_x_out, float* const* grad_feats, float* const* grad_mlp, double* loss_parts,
                    unsigned char* const* touched, void* workspace, size_t workspace_bytes);

// second stage of a fused step (shine_step_support.hip): add `nblocks` per-workgroup partial vectors [PART_STRIDE floats each]
// into the gradient tensors / loss, re-zero the trash rows (set_zero, model/feature_octree.py:78-81)
// cfg->next_draw: pass 1 of the NEXT sorted draw (two-launch form of the sampler: the block sums of its Exp(1) spacings) rides
// on this launch as extra blocks — it depends on nothing but the sampler's stream id
struct Pass1Args {
  double* block_sum;             // null: off
  long long n1;                  // n + 1 spacings
  unsigned long long seed;
  unsigned long long* stream_dev;
  long long* surf_parts;         // cleared here (pass 2 adds to them) or null
  int nblocks;                   // sampler blocks of 1024 draws
};
// cfg->draw_rider (include/shine_hip.h shine_draw_rider): the whole next draw and the next step's zero-fill as extra blocks
struct DrawRiderArgs {
  const double* bs_next;   // pass 1 of the next draw (left by the previous step's launch)
  double* bs_after;        // receives pass 1 of the draw after it
  unsigned long long* state;
  int parity;
  int nblocks;             // sampler blocks of one draw; 0: off
  long long n, pool;
  unsigned long long seed;
  int* idx;
  const unsigned int* surf_bits;
  long long* parts_this;   // cleared (this step's fused kernel has consumed it)
  long long* parts_next;   // receives the next draw's surface counts
  float4* zero_ptr;
  long long zero_n16;
};
__global__ void k_reduce_partials(V1Args a, int nblocks, Pass1Args p1, DrawRiderArgs dr);
int fill_pass1_args(Pass1Args* p1, const shine_step_config* cfg);  // shine_step_support.hip
int fill_draw_rider_args(DrawRiderArgs* dr, const shine_step_config* cfg);
__global__ void k_mark_touched(V1Args a);
int launch_operand_image(const V1Args& a, float* image, hipStream_t st);  // shine_step_support.hip (image: V3_IMAGE_FLOATS floats)

// host: fill everything of V1Args that does not depend on the launch geometry (argument checks included)
inline int fill_step_args(V1Args* a, const shine_tables* t, const shine_step_config* cfg, const float* coord,
                          const float* sdf_label, const float* weight, const int32_t* perm, const int32_t* slots,
                          const int64_t* n_surf, int64_t n, const float* const* feats, const int64_t* rows,
                          const float* const* mlp, float* pred_out, float* grad_x_out, float* const* grad_feats,
                          float* const* grad_mlp, double* loss_parts, unsigned char* const* touched) {
  const bool rec_pool = cfg->sorted_input == 3;  // a pool of 32-byte records behind `coord`
  if (n < 0 || !feats || !rows || !mlp || !grad_feats || (n > 0 && (!coord || (!sdf_label && !rec_pool))))
    return set_error(SHINE_E_INVALID, "shine_train_step: null argument");
  if (rec_pool && (reinterpret_cast<uintptr_t>(coord) & 15))
    return set_error(SHINE_E_INVALID, "shine_train_step: a record pool must be 16-byte aligned");
  if (cfg->n_levels > LCAP) return set_error(SHINE_E_INVALID, "shine_train_step: the MFMA kernels handle up to 4 featured levels");
  const bool weight_inside = rec_pool && cfg->n_levels < 4;  // (the record holds it)
  if (cfg->eikonal_on && ((!weight && !weight_inside) || !n_surf))
    return set_error(SHINE_E_INVALID, "shine_train_step: eikonal needs weight and n_surf");
  if (cfg->loss_weight_on && !weight && !weight_inside)
    return set_error(SHINE_E_INVALID, "shine_train_step: loss_weight_on needs weight");
  LevelSet ls = {};
  int rc = make_level_set(t, cfg, feats, rows, grad_feats, &ls);
  if (rc != SHINE_OK) return rc;
  for (int s = 0; s < cfg->n_levels; ++s) {
    if (!feats[s]) return set_error(SHINE_E_INVALID, "shine_train_step: null feature level");
    // the kernel addresses feature/grad rows with 32-bit float offsets off an SGPR base (one VGPR per address)
    if (ls.lv[s].rows >= (1ll << 29)) return set_error(SHINE_E_INVALID, "shine_train_step: level exceeds 2^29 rows");
    a->lv[s].keys = ls.lv[s].keys;
    a->lv[s].vals = ls.lv[s].vals;
    a->lv[s].feat = ls.lv[s].feat;
    a->lv[s].grad = ls.lv[s].grad;
    a->lv[s].shift = ls.lv[s].shift;
    a->lv[s].mask = ls.lv[s].mask;
    a->lv[s].res = ls.lv[s].res;
    a->rows[s] = ls.lv[s].rows;
    a->feat_rw[s] = const_cast<float*>(feats[s]);
    a->touched[s] = touched ? touched[s] : nullptr;
  }
  for (int k = 0; k < 6; ++k) {
    if (!mlp[k]) return set_error(SHINE_E_INVALID, "shine_train_step: null decoder parameter");
    a->mlp[k] = mlp[k];
    if (cfg->decoder_grad_on) {
      if (!grad_mlp || !grad_mlp[k]) return set_error(SHINE_E_INVALID, "shine_train_step: null decoder grad");
      a->grad_mlp[k] = grad_mlp[k];
    }
  }
  a->coord = coord;
  a->label = sdf_label;
  a->weight = weight;
  a->perm = perm;
  a->slots = slots;
  a->n_surf = cfg->eikonal_on ? reinterpret_cast<const long long*>(n_surf) : nullptr;
  if (cfg->n_surf_parts > 64) return set_error(SHINE_E_INVALID, "shine_train_step: at most 64 n_surf parts");
  a->n_surf_parts = cfg->n_surf_parts > 1 ? cfg->n_surf_parts : 1;
  a->defer_reduce = cfg->defer_reduce ? 1 : 0;
  a->pred = pred_out;
  a->grad_x = grad_x_out;
  a->loss_parts = loss_parts;
  a->n = n;
  a->n_levels = cfg->n_levels;
  a->reduction_sum = cfg->reduction_sum;
  a->decoder_grad_on = cfg->decoder_grad_on;
  a->poly = cfg->poly_int_on;
  a->weighted = cfg->loss_weight_on ? 1 : 0;
  a->pool_mode = cfg->sorted_input == 2 ? 1 : rec_pool ? 2 : 0;
  if (a->pool_mode && !perm) return set_error(SHINE_E_INVALID, "shine_train_step: pool mode needs the sample indices in perm");
  a->ablate = cfg->kernel_variant >> 8;
  a->sigma = cfg->sigma;
  a->inv_n = (float)cfg->inv_n;
  a->weight_e = cfg->weight_e;
  a->adam_state = reinterpret_cast<long long*>(cfg->adam_state);
  a->adam_b1 = cfg->adam_beta1;
  a->adam_b2 = cfg->adam_beta2;
  a->zero_f64 = cfg->zero_f64;
  return SHINE_OK;
}

}  // namespace shine
