// shine_graph.hip — the training iteration as a HIP graph BUILT by the library and RE-BOUND in place.
//
// At the reference's batch size (4096: every shipped yaml) an iteration is two small launches — the fused step and the
// iteration tail (shine_train_step with cfg->defer_reduce, shine_finish_iteration) — and what bounds the loop is the host
// and the idle time around the launches, so the loop is replayed from a HIP graph (shine_batch.py:105-210,
// shine_incre.py:114-181).  Round 3 captured that graph from the stream with torch: incremental mapping re-allocates the
// feature tables and re-creates the optimiser every frame (model/feature_octree.py:147-160, shine_incre.py:107-109), so the
// graph was re-captured and re-instantiated every frame (~0.4 ms of a 3.9 ms frame, and the reason why only 2 iterations per
// graph paid).  Here the graph is a chain of `unroll` x {step node, tail node} created once with hipGraphAddKernelNode from the
// very launches the two entry points would make (prepare_step_v3 / prepare_finish), and a new frame only REWRITES the nodes'
// kernel parameters in the instantiated graph (hipGraphExecKernelNodeSetParams): no capture, no instantiation, and all copies
// of the iteration share one parameter block, because what changes from one iteration to the next — the sampler's stream id,
// Adam's step count — lives in device memory and is advanced by the kernels themselves.
#include <cstdlib>
#include <vector>

#include "shine_finish_args.hpp"
#include "shine_tile16.hpp"

struct shine_iter_graph {
  int unroll = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  std::vector<hipGraphNode_t> step_nodes, fin_nodes;
  shine::StepLaunch step = {};
  shine::FinLaunch fin = {};
  bool have_step = false, have_fin = false, dirty = false;
  const void* built_step_fn = nullptr;
  const void* built_fin_fn = nullptr;
  long long commits = 0, builds = 0;
  // recorded behind the last replay: commit() rewrites the instantiated graph's kernel arguments, and HIP does not say that a
  // replay still queued keeps the old ones (this runtime keeps a graph's kernel arguments in memory the exec owns)
  hipEvent_t last_replay = nullptr;
  bool replayed = false;
  // the decoder's MFMA operand image for the small-batch builds of the step (V1Args::op_image): built at the head of every
  // launch() from the decoder as it is then, kept current by the graph's own tail nodes.  The CALLER's buffer
  // (shine_iter_graph_set_operand_image: the library allocates no device memory for a graph); null: the step builds its operands
  // per workgroup
  float* image = nullptr;
};

namespace shine {

static void graph_drop(shine_iter_graph* g) {
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  g->exec = nullptr;
  g->graph = nullptr;
  g->step_nodes.clear();
  g->fin_nodes.clear();
}

static hipKernelNodeParams step_params(shine_iter_graph* g, void** kp) {
  hipKernelNodeParams p = {};
  kp[0] = &g->step.a;
  p.func = const_cast<void*>(g->step.fn);
  p.gridDim = g->step.grid;
  p.blockDim = g->step.block;
  p.sharedMemBytes = 0;
  p.kernelParams = kp;
  p.extra = nullptr;
  return p;
}

static hipKernelNodeParams fin_params(shine_iter_graph* g, void** kp) {
  hipKernelNodeParams p = {};
  kp[0] = &g->fin.a;
  kp[1] = &g->fin.fb;
  kp[2] = &g->fin.db;
  p.func = const_cast<void*>(g->fin.fn);
  p.gridDim = g->fin.grid;
  p.blockDim = g->fin.block;
  p.sharedMemBytes = 0;
  p.kernelParams = kp;
  p.extra = nullptr;
  return p;
}

static int graph_build(shine_iter_graph* g) {
  graph_drop(g);
  SHINE_HIP_CHECK(hipGraphCreate(&g->graph, 0));
  void* skp[1];
  void* fkp[3];
  hipKernelNodeParams sp = step_params(g, skp), fp = fin_params(g, fkp);
  hipGraphNode_t prev = nullptr;
  for (int u = 0; u < g->unroll; ++u) {
    hipGraphNode_t sn = nullptr, fn = nullptr;
    SHINE_HIP_CHECK(hipGraphAddKernelNode(&sn, g->graph, prev ? &prev : nullptr, prev ? 1 : 0, &sp));
    SHINE_HIP_CHECK(hipGraphAddKernelNode(&fn, g->graph, &sn, 1, &fp));
    g->step_nodes.push_back(sn);
    g->fin_nodes.push_back(fn);
    prev = fn;
  }
  SHINE_HIP_CHECK(hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0));
  g->built_step_fn = g->step.fn;
  g->built_fin_fn = g->fin.fn;
  ++g->builds;
  return SHINE_OK;
}

}  // namespace shine

using namespace shine;

extern "C" int shine_iter_graph_create(int32_t unroll, shine_iter_graph** out) {
  if (!out || unroll < 1 || unroll > 256) return set_error(SHINE_E_INVALID, "shine_iter_graph_create: 1 <= unroll <= 256");
  shine_iter_graph* g = new (std::nothrow) shine_iter_graph();
  if (!g) return set_error(SHINE_E_NOMEM, "shine_iter_graph_create: out of host memory");
  g->unroll = unroll;
  *out = g;
  return SHINE_OK;
}

extern "C" int shine_iter_graph_destroy(shine_iter_graph* g) {
  if (!g) return SHINE_OK;
  graph_drop(g);
  if (g->last_replay) (void)hipEventDestroy(g->last_replay);
  delete g;
  return SHINE_OK;
}

extern "C" int shine_iter_graph_set_step(shine_iter_graph* g, const shine_tables* t, const shine_step_config* cfg,
                                         const float* coord, const float* sdf_label, const float* weight, const int32_t* perm,
                                         const int32_t* slots, const int64_t* n_surf, int64_t n, const float* const* feats,
                                         const int64_t* rows, const float* const* mlp, float* pred_out, float* grad_x_out,
                                         float* const* grad_feats, float* const* grad_mlp, double* loss_parts,
                                         unsigned char* const* touched, void* workspace, size_t workspace_bytes) {
  if (!g || !cfg) return set_error(SHINE_E_INVALID, "shine_iter_graph_set_step: null argument");
  if (!cfg->defer_reduce || ((cfg->kernel_variant & 0xff) != 0 && (cfg->kernel_variant & 0xff) != 4) || n < 1)
    return set_error(SHINE_E_INVALID, "shine_iter_graph_set_step: the graph's iteration is {fused step with cfg->defer_reduce, "
                                      "shine_finish_iteration} on a non-empty planned / pool batch");
  StepLaunch sl;
  int rc = prepare_step_v3(&sl, t, cfg, coord, sdf_label, weight, perm, slots, n_surf, n, feats, rows, mlp, pred_out,
                           grad_x_out, grad_feats, grad_mlp, loss_parts, touched, workspace, workspace_bytes);
  if (rc != SHINE_OK) return rc;
  if (sl.mark_pass || sl.a.prof)
    return set_error(SHINE_E_INVALID, "shine_iter_graph_set_step: touched flags need a gradient table on every level (the "
                                      "marking pass in front of the step is not part of the graph); no profiling build");
  // (SHINE_NO_OPERAND_IMAGE=1 in the environment switches it off: same-box A/B, profiles/r04_ab_experiments.txt block 14)
  static const bool image_off = std::getenv("SHINE_NO_OPERAND_IMAGE") != nullptr;
  if (sl.block.x == 256 && !image_off && g->image)  // a small batch (4-wave workgroups): the step copies the operand image
    sl.a.op_image = g->image;
  g->step = sl;
  g->have_step = true;
  g->dirty = true;
  return SHINE_OK;
}

extern "C" int shine_iter_graph_operand_image_floats(void) { return V3_IMAGE_FLOATS; }

extern "C" int shine_iter_graph_set_operand_image(shine_iter_graph* g, float* image) {
  if (!g) return set_error(SHINE_E_INVALID, "shine_iter_graph_set_operand_image: null graph");
  if (image && ((size_t)image & 15)) return set_error(SHINE_E_INVALID, "shine_iter_graph_set_operand_image: 16-byte aligned buffer");
  g->image = image;
  g->dirty = g->have_step;  // (a bound step names the old buffer: set_step again)
  return SHINE_OK;
}

extern "C" int shine_iter_graph_set_finish(shine_iter_graph* g, const shine_step_config* cfg, int64_t n, const void* workspace,
                                           const int64_t* n_surf, double* loss_parts, const float* const* feats_last,
                                           const float* const* importance, unsigned char* const* touched,
                                           const int32_t* grad_on, float lambda_forget, double* reg_out, int32_t n_tensors,
                                           float* const* params, float* const* grads, float* const* exp_avg,
                                           float* const* exp_avg_sq, const int64_t* numel, const float* lr_dev,
                                           const int32_t* lr_index, const float* weight_decay, float beta1, float beta2,
                                           float eps, const int64_t* step_state, const shine_next_draw* next_draw,
                                           int32_t active_rows) {
  if (!g) return set_error(SHINE_E_INVALID, "shine_iter_graph_set_finish: null graph");
  FinLaunch fl;
  int rc = prepare_finish(&fl, cfg, n, workspace, n_surf, loss_parts, feats_last, importance, touched, grad_on, lambda_forget,
                          reg_out, n_tensors, params, grads, exp_avg, exp_avg_sq, numel, lr_dev, lr_index, weight_decay, beta1,
                          beta2, eps, step_state, next_draw, active_rows);
  if (rc != SHINE_OK) return rc;
  g->fin = fl;
  g->have_fin = true;
  g->dirty = true;
  return SHINE_OK;
}

extern "C" int shine_iter_graph_commit(shine_iter_graph* g) {
  if (!g || !g->have_step || !g->have_fin)
    return set_error(SHINE_E_STATE, "shine_iter_graph_commit: set_step and set_finish first");
  if (g->fin.a.partials != g->step.a.partials || g->fin.a.nblocks != g->step.blocks)
    return set_error(SHINE_E_INVALID, "shine_iter_graph_commit: the tail must consume the step's workspace (same buffer, same batch size)");
  g->fin.a.op_image = g->step.a.op_image ? g->image : nullptr;  // (the tail keeps current what the step copies)
  ++g->commits;
  g->dirty = false;
  if (g->replayed) {
    // waits for the last replay only, not for what the caller has queued since (the frame's octree update): the host keeps
    // running ahead of the device
    SHINE_HIP_CHECK(hipEventSynchronize(g->last_replay));
    g->replayed = false;
  }
  if (g->exec && g->built_step_fn == g->step.fn && g->built_fin_fn == g->fin.fn) {
    void* skp[1];
    void* fkp[3];
    hipKernelNodeParams sp = step_params(g, skp), fp = fin_params(g, fkp);
    bool ok = true;
    for (int u = 0; u < g->unroll && ok; ++u) {
      ok = hipGraphExecKernelNodeSetParams(g->exec, g->step_nodes[u], &sp) == hipSuccess &&
           hipGraphExecKernelNodeSetParams(g->exec, g->fin_nodes[u], &fp) == hipSuccess;
    }
    if (ok) return SHINE_OK;
    (void)hipGetLastError();  // an update this runtime refuses: build the graph afresh
  }
  return graph_build(g);
}

extern "C" int shine_iter_graph_launch(shine_iter_graph* g, int32_t replays, void* stream) {
  if (!g || !g->exec || g->dirty) return set_error(SHINE_E_STATE, "shine_iter_graph_launch: commit first");
  if (replays < 1) return SHINE_OK;
  if (!g->last_replay) SHINE_HIP_CHECK(hipEventCreateWithFlags(&g->last_replay, hipEventDisableTiming));
  if (g->step.a.op_image) {  // the decoder may have been written by anybody since the last replay: the image starts from what it is now
    const int rc = launch_operand_image(g->step.a, g->image, (hipStream_t)stream);
    if (rc != SHINE_OK) return rc;
  }
  for (int r = 0; r < replays; ++r) SHINE_HIP_CHECK(hipGraphLaunch(g->exec, (hipStream_t)stream));
  SHINE_HIP_CHECK(hipEventRecord(g->last_replay, (hipStream_t)stream));
  g->replayed = true;
  return SHINE_OK;
}

extern "C" int shine_iter_graph_stats(const shine_iter_graph* g, int64_t* commits, int64_t* builds) {
  if (!g) return set_error(SHINE_E_INVALID, "shine_iter_graph_stats: null graph");
  if (commits) *commits = g->commits;
  if (builds) *builds = g->builds;
  return SHINE_OK;
}
