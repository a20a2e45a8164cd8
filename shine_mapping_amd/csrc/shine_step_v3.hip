// shine_step_v3.hip — the fused SHINE training step, 16-point tiles, query lane = (point, LEVEL).
//
//   query    FeatureOctree.query_feature   model/feature_octree.py:199-244
//   decode   Decoder.sdf                   model/decoder.py:49-63
//   loss     sdf_bce_loss                  utils/loss.py:17-24
//   backward cur_loss.backward()           shine_batch.py:208-209 (closed form, SURVEY.md §8a math contract)
//
// One wave walks a 16-point tile of the node-ordered stream through every phase.  What the round-2 measurements decided
// (profiles/r02_*): ~1100 VALU + ~600 scalar instructions per tile next to 68 MFMAs in the predecessor (lane = point x corner
// pair), and — the hardware fact behind the design — exact-fp32 MFMA and the VALU work of another wave do NOT overlap on a
// SIMD (tools/ubench/mfma_valu_overlap.hip; SQ_VALU_MFMA_COEXEC_CYCLES = 0): the decoder's matrix work and every vector
// instruction queue for one fp32 datapath.  So this kernel removes instructions and trades waves for registers:
//   * query: lane (pt, g) owns LEVEL g of point pt.  The per-level work — hash slot, node-run masks, smooth-step weights,
//     offsets — is done ONCE per wave instruction instead of once per level; one slot / eight ids per lane; the node-run
//     masks of all four levels come out of ONE 64-bit ballot; the run carry is a DPP row broadcast.  Each lane gathers the
//     eight 32-B corner rows of its level (sixteen 16-B loads, four corners in flight at a time) and sums them with its
//     eight weights; a reduce-scatter over g (v_permlane32_swap / v_permlane16_swap, gfx950) leaves features (2g, 2g+1) of
//     the point in lane g — the B operand of layer 1.  Measured: 1084 -> 744 VALU instructions per tile (SQ_INSTS_VALU);
//   * decoder: exact-fp32 v_mfma_f32_16x16x4_f32 chains whose accumulators ARE the next B operands (the k-order of each
//     product is permuted to the accumulator row order of the previous one); weight grads through [32][20] LDS transposes
//     into MFMA accumulators that live in registers for the whole kernel;
//   * loss: hardware transcendentals (v_exp_f32 / v_rcp_f32 / v_log_f32) instead of libm forms: ~25 VALU, not ~190;
//   * 8 waves per CU (2 per SIMD, <= 256 VGPRs) instead of 12: 69 vs 74.5 us — more waves do not help this kernel;
//   * eikonal build (EIK): the closed-form chain of SURVEY.md §8a on the same tiles, 240 VGPRs, no spills;
//   * scatter: lane = (corner, feature), run-length walk, one 64-lane atomic per node run; the staged operands of level
//     s + 1 are requested before level s is walked;
//   * flush: one partial vector per workgroup, added up by k_reduce_partials (shine_step_support.hip).
// Round 3 (profiles/r03_ab_experiments.txt): role-specialised waves (check/shine_step_v5.hip), a prefix-sum scatter, interleaved
// and claimed tile chunks were all built and measured against this kernel — none is faster; the slowest waves are the ones whose
// tile range holds 4-6 x the mean number of node runs (tools/run_stats.py).
// Planned or pool batches only (the hash slots come with the batch): the Python layer plans every batch that arrives without
// an order (shine_plan_batch), the C entry point refuses one.
#include "shine_step_body.hpp"

namespace shine {

// one launch = one step: every workgroup runs its share of the batch (shine_step_body.hpp)
template <int L, int WAVES, bool EIK, bool PROF, bool EXT = false, bool MARK = false, bool FAR = false>
__global__ __launch_bounds__(WAVES * 64, WAVES == V3_BIG ? V3_BIG / 4 : 2) void k_step_v3(V1Args a) {
  __shared__ StepShared<WAVES, FAR> sm;
  step_body<L, WAVES, EIK, PROF, EXT, MARK, false, FAR>(a, sm, (int)blockIdx.x, (int)gridDim.x);
}

// x[l] + y[l ^ 32] style exchanges through v_permlane32_swap / v_permlane16_swap: pins the lane maps xsum32 / xsum16 assume
__global__ void k_selftest_permlane(const float* x, const float* y, float* o32, float* o16) {
  const int lane = threadIdx.x;
  o32[lane] = xsum32(x[lane], y[lane]);
  o16[lane] = xsum16(x[lane], y[lane]);
}

V2Geometry v3_geometry(long long n, int ablate) {
  V2Geometry g;
  long long tiles = (n + V3_TP - 1) / V3_TP;
  if (tiles < 1) tiles = 1;
  g.tiles = tiles;
  g.wg_waves = tiles < 2048 ? 4 : V3_BIG;  // small batches: 4-wave workgroups spread over more CUs
  const long long max_waves = 256 * V3_BIG;  // resident waves of the full-chip launch
  const long long waves = tiles < max_waves ? tiles : max_waves;
  g.blocks = (waves + g.wg_waves - 1) / g.wg_waves;
  // Round 6 (measured: tools/small_batch_geometry.py, profiles/r06_small_batch_geometry.txt): up to 512
  // tiles as ONE tile per workgroup — one active wave on every CU instead of four on a quarter of the chip — takes the kernel
  // alone from 11.9 to 10.7 us at N = 4096 (13.6 -> 12.5 at 8192, nothing from 1024 tiles on), but every workgroup leaves a
  // partial vector and the iteration's tail (k_finish) then adds up 256 of them instead of 64: the whole graphed iteration
  // 38.6 -> 43.0 us, ncd-incre 610 -> 518 frames/s.  So the product keeps four tiles per workgroup; kernel_variant bit 0x8000
  // asks for the other form (measurement).
  if (tiles <= 512 && (ablate & 128)) g.blocks = tiles;
  g.waves = g.blocks * g.wg_waves;  // every launched wave takes its share (a wave beyond the tile count gets none)
  g.chunk = 0;
  return g;
}

long long v3_lds_bytes(int wg_waves) {
  return (long long)sizeof(float) * (V3_OPTOTAL + 100 + (long long)wg_waves * V3_WAVE_FLOATS) +
         4 * sizeof(double);
}

// the instantiation a launch uses: {levels, workgroup shape, eikonal term, profiling stamps, external delta, touched-row marks,
// FAR = the build for tables beyond the Infinity Cache (full-chip launches of 3- and 4-level trees)}
template <int L, bool EIK, bool EXT, bool MARK>
static const void* step_fn_l(int wg_waves, bool prof, bool far) {
#if SHINE_V3_PROFBUILD  // measurement builds only (tools/mk_variant.py -DSHINE_V3_PROFBUILD=1): the per-wave phase counters
  if (prof && !EXT && !MARK) {
    if constexpr (L >= 3)
      if (far && wg_waves == V3_BIG) return (const void*)k_step_v3<L, V3_BIG, EIK, true, false, false, true>;
    return wg_waves == V3_BIG ? (const void*)k_step_v3<L, V3_BIG, EIK, true> : (const void*)k_step_v3<L, 4, EIK, true>;
  }
#endif
  (void)prof;
  if constexpr (L >= 3 && !EXT)
    if (far && wg_waves == V3_BIG) return (const void*)k_step_v3<L, V3_BIG, EIK, false, false, MARK, true>;
  return wg_waves == V3_BIG ? (const void*)k_step_v3<L, V3_BIG, EIK, false, EXT, MARK>
                            : (const void*)k_step_v3<L, 4, EIK, false, EXT, MARK>;
}
template <bool EIK, bool EXT, bool MARK>
static const void* step_fn(int levels, int wg_waves, bool prof, bool far = false) {
  switch (levels) {
    case 1: return step_fn_l<1, EIK, EXT, MARK>(wg_waves, prof, far);
    case 2: return step_fn_l<2, EIK, EXT, MARK>(wg_waves, prof, far);
    case 3: return step_fn_l<3, EIK, EXT, MARK>(wg_waves, prof, far);
    default: return step_fn_l<4, EIK, EXT, MARK>(wg_waves, prof, far);
  }
}

// FAR or not: by the size of the feature tables (their gradient tables are as large again, the node records as well) against the
// 256 MiB Infinity Cache; kernel_variant's low byte overrides (5: FAR, 6: never) for A/B measurements and the parity tests
constexpr long long FAR_TABLE_BYTES = 256ll << 20;
static bool far_regime(const shine_step_config* cfg, const int64_t* rows) {
  const int v = cfg->kernel_variant & 0xff;
  if (v == 6) return false;
  long long bytes = 0;
  for (int s = 0; s < cfg->n_levels; ++s) {
    if (rows[s] + 1 >= (1ll << V3_ROW_BITS)) return false;  // (a staged corner id of the far build carries 6 lattice bits)
    bytes += (rows[s] + 1) * (long long)(F * sizeof(float));
  }
  return v == 5 || bytes > FAR_TABLE_BYTES;
}

// Everything of a fused-step launch but the launch itself: argument block, kernel instantiation, geometry.  Used by
// shine_train_step_v3 (launches it) and by the iteration graph (shine_graph.hip: a kernel node with these parameters).
int prepare_step_v3(StepLaunch* out, const shine_tables* t, const shine_step_config* cfg, const float* coord,
                    const float* sdf_label, const float* weight, const int32_t* perm, const int32_t* slots,
                    const int64_t* n_surf, int64_t n, const float* const* feats, const int64_t* rows, const float* const* mlp,
                    float* pred_out, float* grad_x_out, float* const* grad_feats, float* const* grad_mlp, double* loss_parts,
                    unsigned char* const* touched, void* workspace, size_t workspace_bytes) {
  if (!slots && cfg->sorted_input != 3)  // (a record pool carries its slots)
    return set_error(SHINE_E_INVALID, "shine_train_step_v3: needs a planned batch (slots)");
  if (cfg->sorted_input == 2)
    return set_error(SHINE_E_INVALID, "shine_train_step_v3: a pool batch is drawn from a RECORD pool (sorted_input 3: one 32-byte "
                                      "record per sample); the array form (2) is the importance sweep's");
  V1Args& a = out->a;
  a = V1Args{};
  int rc = fill_step_args(&a, t, cfg, coord, sdf_label, weight, perm, slots, n_surf, n, feats, rows, mlp, pred_out,
                          grad_x_out, grad_feats, grad_mlp, loss_parts, touched);
  if (rc != SHINE_OK) return rc;
  out->fn = nullptr;
  if (n == 0) return SHINE_OK;
  V2Geometry g = v3_geometry(n, a.ablate);
  if (a.ablate & 64) {  // kernel_variant bit 0x4000: the deterministic (single-wave) launch of the test suite
    g.blocks = 1;
    g.wg_waves = 4;
    g.waves = 4;
  }
  a.tiles = g.tiles;
  a.waves_total = g.waves;
  a.prof = g_prof_buffer;
  const size_t need = (size_t)g.blocks * PART_STRIDE * sizeof(float);
  if (!workspace || workspace_bytes < need)
    return set_error(SHINE_E_INVALID, "shine_train_step_v3: workspace too small (shine_train_step_workspace_bytes)");
  a.partials = (float*)workspace;
  // touched-row flags: set by the scatter of the MARK build (steps whose every level has a gradient table: a level without
  // one is not walked), by a marking pass in front of the step otherwise
  bool mark_in_kernel = touched && !a.prof;
  for (int s = 0; s < cfg->n_levels; ++s) mark_in_kernel = mark_in_kernel && a.lv[s].grad != nullptr;
  out->mark_pass = touched && !mark_in_kernel;
  const bool prof = a.prof != nullptr;
  const bool far = far_regime(cfg, rows);
  if (mark_in_kernel)
    out->fn = cfg->eikonal_on ? step_fn<true, false, true>(cfg->n_levels, g.wg_waves, prof, far)
                              : step_fn<false, false, true>(cfg->n_levels, g.wg_waves, prof, far);
  else
    out->fn = cfg->eikonal_on ? step_fn<true, false, false>(cfg->n_levels, g.wg_waves, prof, far)
                              : step_fn<false, false, false>(cfg->n_levels, g.wg_waves, prof, far);
  out->grid = dim3((unsigned)g.blocks);
  out->block = dim3((unsigned)(g.wg_waves * 64));
  out->blocks = (int)g.blocks;
  return SHINE_OK;
}

}  // namespace shine

using namespace shine;

extern "C" int shine_selftest_permlane(const float* x, const float* y, float* o32, float* o16, void* stream) {
  if (!x || !y || !o32 || !o16) return set_error(SHINE_E_INVALID, "shine_selftest_permlane: null argument");
  hipLaunchKernelGGL(k_selftest_permlane, dim3(1), dim3(64), 0, (hipStream_t)stream, x, y, o32, o16);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_train_step_regime(const shine_step_config* cfg, const int64_t* rows, int64_t n, int32_t* far_out) {
  if (!cfg || !rows || !far_out) return set_error(SHINE_E_INVALID, "shine_train_step_regime: null argument");
  if (cfg->n_levels < 1 || cfg->n_levels > LCAP) return set_error(SHINE_E_INVALID, "shine_train_step_regime: 1..4 featured levels");
  // (the far build exists for full-chip launches of 3- and 4-level trees: step_fn_l)
  *far_out = far_regime(cfg, rows) && cfg->n_levels >= 3 && v3_geometry(n > 0 ? n : 1).wg_waves == V3_BIG ? 1 : 0;
  return SHINE_OK;
}

// the fused step behind shine_train_step (include/shine_hip.h): planned / pool batches, needs the workspace
extern "C" int shine_train_step_v3(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                                   const float* sdf_label, const float* weight, const int32_t* perm,
                                   const int32_t* slots, const int64_t* n_surf, int64_t n, const float* const* feats,
                                   const int64_t* rows, const float* const* mlp, float* pred_out, float* grad_x_out,
                                   float* const* grad_feats, float* const* grad_mlp, double* loss_parts,
                                   unsigned char* const* touched, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  StepLaunch sl;
  int rc = prepare_step_v3(&sl, t, cfg, coord, sdf_label, weight, perm, slots, n_surf, n, feats, rows, mlp, pred_out,
                           grad_x_out, grad_feats, grad_mlp, loss_parts, touched, workspace, workspace_bytes);
  if (rc != SHINE_OK) return rc;
  if (!sl.fn) return SHINE_OK;  // (empty batch)
  hipStream_t st = (hipStream_t)stream;
  V1Args& a = sl.a;
  if (sl.mark_pass) {
    hipLaunchKernelGGL(k_mark_touched, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  void* params[] = {&a};
  SHINE_HIP_CHECK(hipLaunchKernel(sl.fn, sl.grid, sl.block, params, 0, st));
  if (!(a.ablate & 32) && !a.defer_reduce) {  // (ablate bit 32: measurement only — time the dominant kernel by itself)
    Pass1Args p1;  // cfg->next_draw: pass 1 of the next sorted draw as extra blocks of this launch (4 sampler blocks each)
    rc = fill_pass1_args(&p1, cfg);
    if (rc != SHINE_OK) return rc;
    DrawRiderArgs dr;  // cfg->draw_rider: the whole next draw + the next step's zero-fill (2 x 4 sampler blocks per extra block)
    rc = fill_draw_rider_args(&dr, cfg);
    if (rc != SHINE_OK) return rc;
    hipLaunchKernelGGL(k_reduce_partials,
                       dim3((unsigned)((PART_FLOATS + 63) / 64 + (p1.nblocks + 3) / 4 + 2 * ((dr.nblocks + 3) / 4))), dim3(1024), 0,
                       st, a, sl.blocks, p1, dr);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  return SHINE_OK;
}

// Tier A, fused: backward of {FeatureOctree.query_feature -> Decoder.sdf} (model/feature_octree.py:237-244,
// model/decoder.py:49-63) for a given d loss / d pred — what autograd derives for cur_loss.backward()
// (shine_batch.py:208-209) through those two calls, as ONE fused launch (+ the partial-sum reduction): decoder backward,
// decoder weight grads, interpolation backward with the run-merged scatter.  Planned batch (perm, slots from
// shine_plan_batch); grad_pred is indexed like the batch.  grad_feats / grad_mlp are ACCUMULATED INTO.
// grad_g (or NULL) = d loss / d g [n, 3] for g = get_gradient(coord, pred) (utils/tools.py:175-185, WITHOUT the sigma factor the
// drivers multiply in afterwards, cfg->sigma = 1): the eikonal build of the same kernel backpropagates both chains in the one
// launch — what autograd derives through get_gradient(create_graph=True) as a second, double-backward pass.
extern "C" int shine_interp_sdf_backward(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                                         const int32_t* perm, const int32_t* slots, const float* grad_pred,
                                         const float* grad_g, int64_t n,
                                         const float* const* feats, const int64_t* rows, const float* const* mlp,
                                         float* const* grad_feats, float* const* grad_mlp, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (!cfg || !perm || !slots || !grad_pred)
    return set_error(SHINE_E_INVALID, "shine_interp_sdf_backward: needs a planned batch (perm, slots) and grad_pred");
  if (cfg->eikonal_on || cfg->loss_weight_on || cfg->sorted_input == 2)
    return set_error(SHINE_E_INVALID, "shine_interp_sdf_backward: a plain planned batch, no loss options (the loss is the caller's: "
                                      "its eikonal part arrives as grad_g)");
  V1Args a = {};
  // (coord doubles as the label pointer: the external-delta build never reads labels)
  int rc = fill_step_args(&a, t, cfg, coord, coord, nullptr, perm, slots, nullptr, n, feats, rows, mlp, nullptr, nullptr,
                          grad_feats, grad_mlp, nullptr, nullptr);
  if (rc != SHINE_OK) return rc;
  if (n == 0) return SHINE_OK;
  a.ext_delta = grad_pred;
  a.ext_q = grad_g;
  a.inv_n = 1.0f;
  V2Geometry g = v3_geometry(n, a.ablate);
  if (a.ablate & 64) {  // kernel_variant bit 0x4000: the deterministic (single-wave) launch of the test suite, as in prepare_step_v3
    g.blocks = 1;
    g.wg_waves = 4;
    g.waves = 4;
  }
  a.tiles = g.tiles;
  a.waves_total = g.waves;
  const size_t need = (size_t)g.blocks * PART_STRIDE * sizeof(float);
  if (!workspace || workspace_bytes < need)
    return set_error(SHINE_E_INVALID, "shine_interp_sdf_backward: workspace too small (shine_train_step_workspace_bytes)");
  a.partials = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  void* params[] = {&a};
  const void* fn = grad_g ? step_fn<true, true, false>(cfg->n_levels, g.wg_waves, false)
                          : step_fn<false, true, false>(cfg->n_levels, g.wg_waves, false);
  SHINE_HIP_CHECK(hipLaunchKernel(fn, dim3((unsigned)g.blocks), dim3((unsigned)(g.wg_waves * 64)), params, 0, st));
  hipLaunchKernelGGL(k_reduce_partials, dim3((PART_FLOATS + 63) / 64), dim3(1024), 0, st, a, (int)g.blocks, Pass1Args{},
                     DrawRiderArgs{});
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}
