/*
 * shine_hip.h — C ABI of libshine_hip.so: the MI355X (gfx950) SHINE SDF training hot path.
 *
 * The reference (PRBonn/SHINE_mapping) has no FFI of its own: its hot path is the Python
 * surface the two drivers import (shine_batch.py:13-20, shine_incre.py:12-20).  Each entry
 * point below names the reference function(s) it replaces; the Python host side in
 * shine_mapping_amd/ binds them with ctypes and re-exposes the reference's class surface.
 *
 * Conventions
 *  - plain pointers and sizes only; every data pointer is DEVICE memory unless it says "host";
 *  - all work is enqueued on the caller's HIP stream (void* = hipStream_t), no host sync inside
 *    the per-iteration calls (shine_query / shine_train_step / shine_interp_*), no hidden
 *    allocation there either (graph-capturable);
 *  - return 0 on success, a negative SHINE_E_* code otherwise (never throws);
 *    shine_error_string() explains it; HIP failures carry hipGetErrorString text;
 *  - featured levels are addressed by "slot" s = 0..L-1, TOP-DOWN like hier_features
 *    (model/feature_octree.py:61-63): slot L-1 is the leaf level (= tree_level_world),
 *    slot s is absolute level  tree_level_world - (L-1-s);
 *  - feature tables are [rows_s + 1, F] fp32 row-major, the LAST row is the reference's
 *    "trash bin" that index -1 addresses (model/feature_octree.py:76-81,205);
 *  - F (feature_dim) = 8 and H (hidden) = 32, 2 hidden layers: the only decoder shape any
 *    shipped config uses (config/ all yamls: feature_dim 8, mlp_hidden_dim 32, mlp_level 2).
 */
#ifndef SHINE_HIP_H_
#define SHINE_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHINE_MAX_LEVELS 8
#define SHINE_FEATURE_DIM 8
#define SHINE_HIDDEN_DIM 32
/* packed decoder parameter count: W1[32,8] b1[32] W2[32,32] b2[32] w3[32] b3[1] */
#define SHINE_MLP_PARAMS 1377

#define SHINE_OK 0
#define SHINE_E_INVALID (-1)   /* bad argument (null pointer, level out of range, unsupported shape) */
#define SHINE_E_HIP (-2)       /* a HIP runtime call failed; see shine_error_string */
#define SHINE_E_NOMEM (-3)     /* device allocation failed */
#define SHINE_E_STATE (-4)     /* table handle in the wrong state (e.g. level never built) */

typedef struct shine_tables shine_tables; /* opaque: device hash tables node-morton -> 8 corner ids */

/* a sorted draw of the NEXT iteration handed to a call of THIS iteration, so that it costs no launch (or one launch less) of
 * its own: what shine_sample_sorted_dev(pool_size, n, seed, stream_state, idx_out, NULL, 0, surf_bits, surf_parts, ...) would
 * launch.  Taken by shine_finish_iteration (whole draw, n < 16 K) and by shine_step_config.next_draw (pass 1 of a large draw). */
#define SHINE_SURF_PARTS 64
typedef struct shine_next_draw {
  int64_t pool_size, n;
  uint64_t seed;
  uint64_t* stream_state;  /* device uint64[4], as shine_sample_sorted_dev */
  int32_t* idx_out;        /* [n] */
  const uint32_t* surf_bits; /* with surf_parts: one bit per pool sample, weight > 0 */
  int64_t* surf_parts;     /* int64[SHINE_SURF_PARTS] or NULL */
  void* workspace;         /* the draw's workspace (shine_sample_sorted_* with workspace == NULL sizes it): used by
                              shine_step_config.next_draw only */
} shine_next_draw;

/* The WHOLE next large sorted draw — and the next step's opt.zero_grad() — on the reduction launch of THIS step
 * (shine_step_config.draw_rider): a batch-mode step is then two launches, the fused kernel and this one.  The sampler's two
 * passes have a grid-wide dependency (pass 2 needs every block sum of pass 1), so they ride one step apart: step i's launch runs
 * pass 2 of draw i + 1 (whose block sums step i - 1's launch left) and pass 1 of draw i + 2.  Everything a step hands to the next
 * is kept twice and addressed by the step's PARITY (steps alternate 0, 1, 0, ...):
 *   state        device uint64[2]: state[parity] = stream id of the draw THIS step used; the launch stores that + 1 into
 *                state[1 - parity] (nothing in the launch reads it)
 *   block_sum    two device double[(n + 1 + 1023) / 1024]: [1 - parity] holds pass 1 of the next draw, [parity] receives pass 1 of the
 *                draw after it
 *   surf_parts   (with surf_bits) two device int64[SHINE_SURF_PARTS]: [parity] — this step's count, consumed by its fused
 *                kernel — is cleared, [1 - parity] receives the next draw's
 *   zero_ptr     the NEXT step's gradient bucket (two buckets alternate: this step's own holds its results), cleared here; or NULL
 * idx_out [n]: the next step's sorted sample indices (the fused kernel of this step has read its own by now).
 * The draws are bit-identical to shine_sample_sorted with the same seed / stream ids.  shine_draw_rider_prime sets the chain up:
 * draw 0 (stream id first_stream_id) into idx_out / surf_parts[0], pass 1 of draw 1 into block_sum[1], state[0], surf_parts[1] = 0;
 * the first step then has parity 0. */
typedef struct shine_draw_rider {
  int64_t pool_size, n;
  uint64_t seed;
  uint64_t* state;
  int32_t parity;
  double* block_sum[2];
  int32_t* idx_out;
  const uint32_t* surf_bits;
  int64_t* surf_parts[2];
  float* zero_ptr;
  int64_t zero_bytes;
} shine_draw_rider;
int shine_draw_rider_prime(const shine_draw_rider* r, uint64_t first_stream_id, void* stream);

/* FeatureOctree.cal_regularization (model/feature_octree.py:246-255) riding on the query's launch (shine_forward, cfg->reg_rider):
 * every row the batch addresses — unique(hierarchical_indices) without -1 — adds sum_f importance[row][f] * (F[row][f] -
 * F_last[row][f])^2 ONCE per launch (a row is claimed by an atomic exchange of its stamp with the launch's epoch).  The VALUE only
 * (config 4 from its second frame on: features_last_frame is an attached clone, its gradient cancels, :160).  All arrays top-down
 * like the feature tables; stamp[s]: device uint32[rows_s + 1], zero-filled when created; epoch: > 0, strictly increasing from
 * launch to launch of one octree; acc: device float[8] (zero-filled when created): this launch accumulates into acc[epoch & 7] and
 * clears acc[(epoch + 1) & 7] for the next one. */
typedef struct shine_reg_rider {
  const float* last[SHINE_MAX_LEVELS];   /* features_last_frame */
  const float* imp[SHINE_MAX_LEVELS];    /* importance_weight */
  uint32_t* stamp[SHINE_MAX_LEVELS];
  uint32_t epoch;
  float* acc;
} shine_reg_rider;

/* scalar configuration of one hot-path call (plain-old-data, passed by pointer from the host) */
typedef struct shine_step_config {
  int32_t n_levels;        /* L = tree_level_feat                      (utils/config.py:78)  */
  int32_t max_level;       /* tree_level_world                         (utils/config.py:77)  */
  int32_t poly_int_on;     /* smooth-step interpolation                (feature_octree.py:176-179) */
  int32_t reduction_sum;   /* 0: "mean", 1: "sum"                      (utils/loss.py:17-24, shine_incre.py:77-78) */
  int32_t eikonal_on;      /* closed-form eikonal term                 (shine_batch.py:141-142,182-185) */
  int32_t decoder_grad_on; /* 0 when the decoder is frozen             (utils/tools.py:188-191) */
  int32_t sorted_input;    /* 0: visit the batch as given; 1: through perm[] (shine_plan_batch / shine_morton_sort);
                              2: POOL mode — coord/label/weight/slots are a node-ordered sample pool, perm[] holds the
                              batch's sorted sample indices (shine_sample_sorted), outputs are written at batch position;
                              3: POOL mode with the pool as ONE 16-byte-aligned 32-byte record per sample — `coord` is the record
                              base: {x, y, z, label | weight, slot[0..L-1], 0...} for n_levels <= 3, {x, y, z, label | slot[0..3]}
                              for n_levels = 4 (`weight` then stays a separate array, needed by eikonal / loss_weight_on only);
                              sdf_label is ignored, `slots` must be non-NULL and is not read (pass the record base) — a drawn
                              sample costs one cache line instead of four (shine_train_step, shine_mark_touched) */
  int32_t kernel_variant;  /* low byte — 0 (or 4): the fused step (shine_step_v3.hip: planned / pool batches, <= 4 featured
                              levels; 5 / 6 force / forbid its build for tables beyond the Infinity Cache, which 0 picks by
                              table size).  The CHECK library (libshine_check.so, tests / tools only) adds 1: the lane-per-point
                              reference kernel (any batch, <= 8 levels) */
  float sigma;             /* sigma_sigmoid = ratio*sigma_m*scale      (shine_batch.py:87) */
  float weight_e;          /* eikonal weight                           (config weight_e) */
  double inv_n;            /* 1/N_global for "mean", 1 for "sum"       */
  int64_t n_global;        /* global batch size (data parallel)        */
  int32_t sort_origin[3];  /* leaf-level voxel coords of the map's bounding-box corner (shine_morton_sort) */
  int32_t sort_bits[3];    /* bits per axis that cover the box; 0 = whole cube (tree_level_world bits)     */
  int32_t loss_weight_on;  /* 1: BCEWithLogitsLoss(weight=|weight|) (utils/loss.py:18-19, shine_batch.py:172-174):
                              every sample's BCE term is multiplied by |weight[i]| (the reduction still divides by N) */
  /* Iteration hooks of shine_train_step (all optional, NULL = off): scalar housekeeping of the calls that FOLLOW the step in
   * an iteration rides on the step's reduction launch instead of costing launches of its own — at the reference's batch
   * size (4096) an iteration is a chain of ~8 small launches and each one costs its run time plus ~2 us of dependency gap. */
  int64_t* adam_state;     /* the shine_adam_step_dev state of the optimiser step that follows: the step counts it and derives
                              the bias corrections; pass zero_grad | 2 to that shine_adam_step_dev call */
  float adam_beta1, adam_beta2;
  double* zero_f64;        /* one device double cleared by the step (shine_regularize's accumulator: out_zeroed = 1 there) */
  int32_t n_surf_parts;    /* shine_train_step's n_surf points at this many (<= 64) int64 partial counts which the kernels add up
                              (what shine_sample_sorted_* writes to surf_parts); 0 / 1 = one count */
  const shine_next_draw* next_draw; /* host pointer or NULL: pass 1 of the NEXT large sorted draw (the block sums of its spacings:
                              n, seed, stream_state, surf_parts and workspace of the record are used) rides on this step's
                              reduction launch; that draw is then completed by shine_sample_sorted_finish (one launch
                              instead of two).  Ignored with defer_reduce. */
  int32_t defer_reduce;    /* 1: shine_train_step launches the fused kernel ONLY and leaves its per-workgroup partial sums
                              (decoder grads, trash-row grads, loss terms) in the workspace: shine_finish_iteration consumes
                              them in the optimiser's launch.  loss_parts is then written by that call, and adam_state /
                              zero_f64 are served by the fused kernel itself. */
  const shine_reg_rider* reg_rider; /* host pointer or NULL: shine_forward also evaluates the regulariser (see shine_reg_rider) */
  const shine_draw_rider* draw_rider; /* host pointer or NULL: shine_train_step's reduction launch draws the next batch and clears
                              the next step's gradient bucket (see shine_draw_rider).  Not with next_draw / defer_reduce. */
} shine_step_config;

/* ---- library ------------------------------------------------------------------------- */
int shine_version(void);
const char* shine_error_string(int code);

/* ---- lookup tables: replace the python dicts nodes_lookup_tables[level]
 *      (model/feature_octree.py:47-52, filled at :162-166, read at :209) ------------------ */
int shine_tables_create(int32_t n_levels, shine_tables** out);
int shine_tables_destroy(shine_tables* t);
/* insert n NEW nodes of one level: keys = kaolin-convention Morton codes (x MSB per triplet),
 * corner_ids[n,8] = feature-row ids in the corner order of interpolat (:186-193).  Grows/rehashes
 * as needed (this call may synchronise the stream; it runs per frame, not per iteration). */
int shine_tables_insert(shine_tables* t, int32_t slot, const int64_t* keys, const int32_t* corner_ids,
                        int64_t n, void* stream);
int shine_tables_stats(const shine_tables* t, int32_t slot, int64_t* capacity, int64_t* count);
/* Device arrays a growth replaced (a rehashed table, an outgrown scratch buffer) are not freed on the spot — launches bound to
 * them may be in flight, and hipFree waits for the whole device — but retired.  shine_tables_retired_bytes reports how much the
 * handle holds that way; shine_tables_trim frees it.  Call trim only when no launch that was given the OLD arrays can still be
 * pending (after a device synchronisation, with every iteration graph re-bound since the growth). */
int shine_tables_retired_bytes(const shine_tables* t, int64_t* bytes);
int shine_tables_trim(shine_tables* t, int64_t* freed_bytes);

/* ---- octree growth on the device: FeatureOctree.update (model/feature_octree.py:114-166), SURVEY.md §8 f-2.
 *      points[n,3] f32 (device, scaled to [-1,1]) = the frame's surface points.  Inserts every node the frame
 *      creates (all featured levels) with its 8 corner ids; corner ids are assigned exactly as the reference does
 *      (new corners in lexicographic (x,y,z) order after the existing rows, :132-151).  Writes to HOST arrays
 *      fresh_counts[L] (new nodes per level, top-down) and added_counts[L] (new feature rows per level): the caller
 *      appends that many rows to hier_features (:139,153).  Synchronises the stream twice (it runs per frame).
 *      The corner tables (corners_lookup_tables, :47-52) live in the handle; shine_tables_insert_corners seeds them
 *      when nodes were inserted with shine_tables_insert (corner_keys = x<<42 | y<<21 | z at that level). ------- */
int shine_tables_grow(shine_tables* t, const shine_step_config* cfg, const float* points, int64_t n,
                      int64_t* fresh_counts, int64_t* added_counts, void* stream);
/* copy out what the last shine_tables_grow added to one level (any pointer may be NULL): node keys [fresh] in
 * insertion (= Morton) order, their corner ids [fresh,8], the new corners' keys [added] in row-id order.
 * Valid until the next shine_tables_grow / shine_tables_rank_nodes on this handle. */
/* shine_tables_grow_fetch_all: the same for all levels at once, one launch: out (device int64[out_words]) receives, level after
 * level (top-down), {node keys [fresh] | corner ids [fresh][8] int32 = 4 words per node | new corner keys [added]};
 * out_words >= sum over levels of 5 * fresh + added (the counts shine_tables_grow returned). */
int shine_tables_grow_fetch_all(const shine_tables* t, int64_t* out, int64_t out_words, void* stream);
int shine_tables_grow_fetch(const shine_tables* t, int32_t slot, int64_t* fresh_keys, int32_t* fresh_ids,
                            int64_t* new_corner_keys, void* stream);
int shine_tables_insert_corners(shine_tables* t, int32_t slot, const int64_t* corner_keys, const int32_t* ids,
                                int64_t n, void* stream);
int shine_tables_corner_count(const shine_tables* t, int32_t slot, int64_t* count);
/* rank every node of every featured level in one Z-order on the device (what shine_tables_set_ranks uploads when the
 * host computes it); n_buckets_out (host, may be NULL) = nodes + 64 miss buckets. */
int shine_tables_rank_nodes(shine_tables* t, int64_t* n_buckets_out, void* stream);

/* ---- FeatureOctree.get_indices (model/feature_octree.py:199-218):
 *      idx_out[i] for i = 0..L-1 BOTTOM-UP (i = 0 leaf) each [N,8] int64, -1 on miss --------- */
int shine_query_indices(const shine_tables* t, const shine_step_config* cfg, const float* coord, int64_t n,
                        int64_t* const* idx_out /* host array of L device pointers */, void* stream);

/* ---- forward only: FeatureOctree.query_feature (:237-244) + Decoder.sdf (model/decoder.py:49-63).
 *      feats: host array of L device pointers (top-down), rows: host array of L row counts
 *      (without the trash row).  mlp: 6 device pointers W1,b1,W2,b2,w3,b3 (host array).
 *      Any of feat_out [N,8], pred_out [N], idx_out (as above), grad_x_out [N,3]
 *      (= d pred / d coord * sigma, utils/tools.py:175-185) may be NULL.
 *      Like query_feature (set_zero, :78-81,238) the call re-zeroes the trash row feats[s][rows[s]] of every level —
 *      the one write through the `feats` pointers. ------------------------------------------------------------- */
int shine_forward(const shine_tables* t, const shine_step_config* cfg, const float* coord, int64_t n,
                  const float* const* feats, const int64_t* rows, const float* const* mlp, float* feat_out,
                  float* pred_out, int64_t* const* idx_out, float* grad_x_out, void* stream);

/* ---- Tier A (strict drop-in): sdf_bce_loss (utils/loss.py:17-24) = BCEWithLogitsLoss(reduction, weight)(pred,
 *      sigmoid(sdf_label / sigma)) and its derivative in ONE launch: *loss_out (device float) = the mean (reduction_sum = 0)
 *      or the sum of the per-sample terms, each multiplied by weight[i] when weight != NULL (loss_weight_on);
 *      dpred_out [n] (or NULL) = d loss / d pred — what autograd's backward of the torch composite produces in three more
 *      launches.  One workgroup up to 16384 points (the reference's batch sizes are a few thousand), a grid of them beyond
 *      (their partial sums meet in fp32 atomics: the last bit of the loss may differ from run to run there). */
int shine_bce_loss(const float* pred, const float* sdf_label, const float* weight, int64_t n, float sigma,
                   int32_t reduction_sum, float* loss_out, float* dpred_out, void* stream);

/* ---- Tier A (strict drop-in): the backward of FeatureOctree.query_feature as autograd derives it from
 *      model/feature_octree.py:222-234, and its own backward (needed by get_gradient(create_graph=True),
 *      utils/tools.py:175-185, when the eikonal term is differentiated, shine_batch.py:182-185).
 *      grad_feat [N,8] = d loss / d feat.  grad_coord_out [N,3] or NULL.  grad_feats[s] [rows_s+1,8] ACCUMULATED
 *      INTO (trash row included) or NULL array / NULL entries.
 *      _backward_backward: gg_coord [N,3] = d loss / d(grad_coord); grad_gfeat_out [N,8] = d loss / d(grad_feat). */
int shine_interp_backward(const shine_tables* t, const shine_step_config* cfg, const float* coord, int64_t n,
                          const float* const* feats, const int64_t* rows, const float* grad_feat,
                          float* grad_coord_out, float* const* grad_feats, void* stream);
int shine_interp_backward_backward(const shine_tables* t, const shine_step_config* cfg, const float* coord, int64_t n,
                                   const float* const* feats, const int64_t* rows, const float* grad_feat,
                                   const float* gg_coord, float* grad_gfeat_out, float* const* grad_feats,
                                   void* stream);

/* ---- Tier A (strict drop-in): Decoder.sdf (model/decoder.py:49-63) as a twice-differentiable op — forward, the
 *      backward autograd derives for cur_loss.backward() (shine_batch.py:208-209), and the backward of that backward,
 *      needed when get_gradient(create_graph=True) (utils/tools.py:175-185) feeds the eikonal term
 *      (shine_batch.py:141-142,182-185).  feat [N,8]; mlp = 6 device pointers W1,b1,W2,b2,w3,b3 (host array).
 *      _backward: grad_pred [N] = d loss / d pred;  grad_feat_out [N,8] (overwritten) or NULL;  grad_mlp: NULL or 6
 *      device pointers ACCUMULATED INTO (shapes of mlp).
 *      _backward_backward: gg_feat [N,8] = d loss / d(grad_feat);  grad_gpred_out [N] (overwritten) or NULL;
 *      grad_mlp: NULL or 6 pointers, W1/W2/w3 accumulated into (the biases receive nothing). ----------------------- */
int shine_mlp_forward(const float* feat, int64_t n, const float* const* mlp, float* pred_out, void* stream);
int shine_mlp_backward(const float* feat, const float* grad_pred, int64_t n, const float* const* mlp,
                       float* grad_feat_out, float* const* grad_mlp, void* stream);
int shine_mlp_backward_backward(const float* feat, const float* grad_pred, const float* gg_feat, int64_t n,
                                const float* const* mlp, float* grad_gpred_out, float* const* grad_mlp, void* stream);

/* ---- Tier A, fused: backward of { FeatureOctree.query_feature (model/feature_octree.py:237-244) -> Decoder.sdf
 *      (model/decoder.py:49-63) } for a given grad_pred [N] = d loss / d pred — what autograd derives for
 *      cur_loss.backward() (shine_batch.py:208-209) through those two calls when the driver is unchanged: decoder
 *      backward, decoder weight grads and the interpolation backward with the run-merged scatter in ONE fused launch
 *      (+ the partial-sum reduction).  The batch must be planned (perm, slots: shine_plan_batch); grad_pred is indexed like
 *      the batch.  grad_feats[s] [rows_s+1, 8] and grad_mlp[6] (NULL entries / cfg->decoder_grad_on = 0: skipped) are
 *      ACCUMULATED INTO.  No loss options: the loss is the caller's (any torch code).
 *      grad_g (NULL, or [N, 3] indexed like the batch) = d loss / d g for g = cfg->sigma * d pred / d coord, i.e. what
 *      get_gradient(coord, pred) returns (utils/tools.py:175-185; the drivers multiply sigma_sigmoid in afterwards, so the
 *      host side passes cfg->sigma = 1): with it the same launch also backpropagates the eikonal chain (shine_batch.py:141-142,
 *      182-185) — autograd's second, double-backward pass through get_gradient(create_graph=True) — in closed form
 *      (SURVEY.md §8a), grad_pred being whatever reaches pred from the rest of the loss. ------------------------------------ */
int shine_interp_sdf_backward(const shine_tables* t, const shine_step_config* cfg, const float* coord, const int32_t* perm,
                              const int32_t* slots, const float* grad_pred, const float* grad_g, int64_t n,
                              const float* const* feats,
                              const int64_t* rows, const float* const* mlp, float* const* grad_feats,
                              float* const* grad_mlp, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused training step: query + decode + sdf_bce_loss (utils/loss.py:17-24) [+ eikonal
 *      (shine_batch.py:182-185)] + the whole backward (shine_batch.py:208-209) in one pass.
 *      Inputs : coord [N,3], sdf_label [N], weight [N] (sign = surface/free, data_sampler.py:102-103),
 *               perm [N] int32 (visiting order) and slots [N,L] int32 (per point IN VISITING ORDER the hash slot of its
 *               node at each level, -1 = miss): both come from shine_plan_batch, or the batch is a pool draw
 *               (sorted_input 2).  A batch without slots is refused (SHINE_E_INVALID) by the product library,
 *               n_surf: device int64 (global #weight>0) or NULL when eikonal off.
 *      Outputs: pred [N]; grad_x [N,3] or NULL; grad_feats[s] [rows_s+1, 8] and grad_mlp[6]
 *               ACCUMULATED INTO (caller zero-fills; matches autograd's dense grads incl. the
 *               trash row); loss_parts: device double[4] OVERWRITTEN with
 *               [0] BCE term (already reduced per cfg), [1] eikonal mean term (unweighted),
 *               [2] number of points processed, [3] [0] + weight_e * [1].
 *               The trash row of every feats[s] is re-zeroed (FeatureOctree.set_zero, :78-81).
 *               touched: NULL, or L device byte arrays [rows_s]: the step sets touched[s][r] = 1 for every row r
 *               that received gradient (= the unique() of hierarchical_indices without -1), for shine_regularize. */
int shine_train_step(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                     const float* sdf_label, const float* weight, const int32_t* perm, const int32_t* slots,
                     const int64_t* n_surf, int64_t n, const float* const* feats, const int64_t* rows,
                     const float* const* mlp,
                     float* pred_out, float* grad_x_out, float* const* grad_feats, float* const* grad_mlp,
                     double* loss_parts, unsigned char* const* touched, void* workspace, size_t workspace_bytes,
                     void* stream);
/* bytes of device scratch shine_train_step wants for a batch of n points (per-workgroup partial sums of
 * the decoder / trash-row gradients, reduced by a second tiny kernel: deterministic, no hot-spot atomics).
 * workspace may be NULL (or too small): the step then falls back to fp32 atomics for those sums. */
size_t shine_train_step_workspace_bytes(const shine_step_config* cfg, int64_t n);

/* measurement aid: what one shine_train_step launch of n points looks like.  out: int64[8] = {workgroups, waves,
 * points per tile, MFMA FLOP issued per tile (padding included), LDS bytes per workgroup, useful decoder FLOP per point
 * (SURVEY.md §8d), 0, 0}.  bench.py derives its `roofline.mfma_*` figures from these instead of hard-coding them. */
int shine_train_step_info(const shine_step_config* cfg, int64_t n, int64_t* out);
/* which build of the fused step a launch of n points on tables of rows[s] rows (trash row excluded) would run: *far_out = 1 for
 * the build for feature tables beyond the 256 MiB Infinity Cache (closed node runs merged in a per-wave corner lattice in LDS
 * before they become atomics), 0 for the cache-resident build.  The choice is by table size; cfg->kernel_variant's low byte 5 / 6 forces one or the
 * other (A/B measurements, parity tests). */
int shine_train_step_regime(const shine_step_config* cfg, const int64_t* rows, int64_t n, int32_t* far_out);
/* the stand-alone form of the `touched` pass of shine_train_step: mark (set to 1) in touched[s][rows_s] every feature row
 * that the n samples address (perm / slots / cfg->sorted_input as for shine_train_step; coord may be NULL when slots are
 * given).  Data-parallel ranks mark the rows of the GLOBAL draw with it (every rank derives the same set without a
 * collective), then exchange only those rows (shine_mapping_amd/dp.py). */
int shine_mark_touched(const shine_tables* t, const shine_step_config* cfg, const float* coord, const int32_t* perm,
                       const int32_t* slots, int64_t n, const int64_t* rows, unsigned char* const* touched, void* stream);

/* ---- Morton ordering of a batch: the new step right after LiDARDataset.get_batch
 *      (dataset/lidar_dataset.py:430-450).  perm_out[N] int32 = argsort of the leaf-level node keys, to be
 *      passed as `perm` to shine_train_step.  Call with workspace == NULL to get the required bytes. --- */
int shine_morton_sort(const shine_step_config* cfg, const float* coord, int64_t n, int32_t* perm_out,
                      void* workspace, size_t* workspace_bytes, void* stream);

/* ---- incremental mapping epilogues (config/ncd/ncd_incre_reg.yaml)
 *      shine_regularize = FeatureOctree.cal_regularization (model/feature_octree.py:246-255) on the rows flagged by the
 *      last shine_train_step: *reg_out = sum importance*(F - F_last)^2 (unweighted; overwritten), and, for levels with
 *      grad_on[s] != 0, grad_feats[s] += 2*lambda*importance*(F - F_last).  grad_on[s] = 0 is the reference's
 *      attached-clone quirk (:160): value only.  Clears the flags — unless keep_flags != 0: the flags are then the optimiser's
 *      sticky active-row flags (shine_adam_step row_flags), only rows with bit 0 set (touched by THIS iteration's step) count,
 *      and the optimiser's launch turns them into "touched earlier".  out_zeroed != 0: *reg_out was already cleared (by the
 *      step's reduction launch, cfg->zero_f64) — no memset launch here.
 *      shine_importance_accumulate = the per-chunk epilogue of cal_feature_importance (utils/incre_learning.py:36-40):
 *      importance += |grad|, grad = 0, importance[trash row] = 0, for one level ([rows+1, 8] tensors). ------------- */
int shine_regularize(int32_t n_levels, const float* const* feats, const float* const* feats_last,
                     const float* const* importance, float* const* grad_feats, unsigned char* const* touched,
                     const int64_t* rows, const int32_t* grad_on, float lambda_forget, double* reg_out,
                     int32_t out_zeroed, int32_t keep_flags, void* stream);
int shine_importance_accumulate(float* importance, float* grad, int64_t rows, void* stream);
/*      shine_append_rows = the feature-side appends of FeatureOctree.update (model/feature_octree.py:147-160) for n_levels
 *      levels that received `added[s]` new corners, ONE launch: feat[s] [old_rows + added + 1, 8] = the old rows without their
 *      trash row, then feature_std * noise[s] (noise[s] = the caller's randn(added + 1, 8): the random stream stays the
 *      caller's) with the last row — the new trash row — zeroed; imp[s] (or NULL) = old_imp[s] without its trash row, then
 *      zeros; last[s] (or NULL) = a copy of feat[s] (features_last_frame).  old_rows[s] excludes the trash row. */
int shine_append_rows(int32_t n_levels, const float* const* old_feat, const float* const* old_imp, const float* const* noise,
                      const int64_t* old_rows, const int64_t* added, float feature_std, float* const* feat, float* const* imp,
                      float* const* last, void* stream);
/*      shine_importance_sweep = the whole of cal_feature_importance's chunk loop (utils/incre_learning.py:27-40) in one
 *      call.  coord / sdf_label / weight (or NULL) / slots: a node-ordered pool as for pool-mode shine_train_step
 *      (cfg->sorted_input = 2, eikonal off, decoder_grad_on = 0); idx: the chunks' members (sorted sample indices into the
 *      pool) stored chunk after chunk; chunk_begin: HOST int64[n_chunks + 1] offsets into idx.  Per chunk: the fused step
 *      with inv_n = 1 / chunk size (or 1 under reduction_sum) into the chunk's OWN gradient tables (the reference sums a
 *      chunk's gradient before the abs, :36-38), then importance[s] += |that gradient| for every level, chunk after chunk,
 *      importance[s][trash row] = 0.  `group` chunks (1..64) are ONE launch of the step and one launch that folds their
 *      tables into importance: scratch = zeroed device memory for `group` sets of {gradient tables, one flag byte per row}
 *      — zero on entry, zero again on return, so it can be kept and re-used — workspace = the steps' partial vectors;
 *      shine_importance_sweep_sizes gives group (64, fewer chunks, or what fits budget_bytes of scratch; 0 = no budget) and both
 *      sizes.  Like the query_feature of every chunk (set_zero, model/feature_octree.py:78-81,238) the call re-zeroes the
 *      trash row of every feats[s] — the one write through the `feats` pointers. */
/*      shine_importance_chunks = the chunks of that loop (chunk c = pool[c * bs * down_rate : (c + 1) * bs * down_rate :
 *      down_rate], :27-31) for a node-ordered pool: perm[j] = ORIGINAL pool index of the sample at sorted position j.  Writes
 *      idx_out (device int32, >= number of kept samples, <= n): the sorted positions of the chunks' members, chunk after chunk,
 *      ascending inside a chunk, and chunk_begin (HOST int64[n_chunks + 1], n_chunks = ceil(n / (bs * down_rate))): what
 *      shine_importance_sweep takes.  workspace == NULL: *workspace_bytes = the size needed. */
int shine_importance_chunks(const int32_t* perm, int64_t n, int64_t bs, int32_t down_rate, int32_t* idx_out,
                            int64_t* chunk_begin, int32_t n_chunks, void* workspace, size_t* workspace_bytes, void* stream);
int shine_importance_sweep_sizes(int32_t n_levels, const int64_t* rows, int32_t n_chunks, int64_t max_chunk,
                                 size_t budget_bytes, int32_t* group_out, size_t* scratch_bytes, size_t* workspace_bytes);
int shine_importance_sweep(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                           const float* sdf_label, const float* weight, const int32_t* idx, const int32_t* slots,
                           const int64_t* chunk_begin, int32_t n_chunks, const float* const* feats, const int64_t* rows,
                           const float* const* mlp, float* const* importance, int32_t group, void* scratch,
                           size_t scratch_bytes, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused dense Adam (next row f-1): opt.step() [+ opt.zero_grad()] of shine_batch.py:208-210 for the optimiser of
 *      setup_optimizer (utils/tools.py:57-83): torch.optim.Adam semantics (betas, eps, L2 weight decay added to the
 *      grad, bias correction with the 1-based `step`), one lr / weight_decay per tensor, every element updated
 *      (dense, like the reference).  n_tensors <= 16; all arrays are host arrays of n_tensors entries.
 *      row_flags (NULL, or a host array of n_tensors device pointers with NULL entries for dense tensors): EXACT active rows
 *      for [rows, 8] feature tables without weight decay — one byte per row, 0 = the row has had no gradient since the
 *      optimiser state was created (m = v = g = 0: torch's Adam leaves it bit for bit unchanged, so it is skipped unread),
 *      bit 0 = touched by this iteration's step (shine_train_step `touched`; this launch rewrites it as 2), 2 = touched
 *      earlier.  The caller clears flags and state together. ---------------------------------------------------------- */
int shine_adam_step(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int64_t* numel, const float* lr, const float* weight_decay,
                    float beta1, float beta2, float eps, int64_t step, int32_t zero_grad,
                    unsigned char* const* row_flags, void* stream);

/* ---- Batch plan: order a batch by octree node (counting sort) and remember every point's hash slots.
 *      shine_tables_set_ranks: per level, the rank of every node in ONE Z-order over all featured levels (a parent's
 *      own bucket right after its children's); n_buckets = number of nodes of all levels + 64 (miss buckets); call for every level
 *      after the tree grew, the leaf level (slot L-1) last.  keys/ranks: device arrays of n entries.
 *      shine_plan_batch: perm_out [N] int32 (visiting order) and slots_out [N,L] int32 (in visiting order) to pass
 *      to shine_train_step.  zero_ptr/zero_bytes (16-B aligned, may be NULL/0): a buffer cleared in the same pass —
 *      the flat gradient bucket, i.e. opt.zero_grad() for the fused step.  workspace == NULL returns the required
 *      bytes.  Replaces shine_morton_sort on the training path: ~3 small launches, and the fused kernel no longer
 *      hashes or probes.  Batches of <= 16384 points (the reference's batch size is 4096) are NOT reordered — perm_out is the
 *      identity, one launch: the fused step does not see the order at that size — unless cfg->kernel_variant has bit 0x800 set
 *      (a sample POOL that is planned once and drawn from many times wants the node order at any size). -------------------- */
int shine_tables_set_ranks(shine_tables* t, int32_t slot, const int64_t* keys, const int32_t* ranks, int64_t n,
                           int64_t n_buckets, void* stream);
int shine_plan_batch(const shine_tables* t, const shine_step_config* cfg, const float* coord, int64_t n,
                     int32_t* perm_out, int32_t* slots_out, void* zero_ptr, size_t zero_bytes, void* workspace,
                     size_t* workspace_bytes, void* stream);

/* ---- sampler + order fusion (next row f-3): LiDARDataset.get_batch (dataset/lidar_dataset.py:430-450) as SORTED i.i.d.
 *      uniform indices (exact order statistics via exponential spacings: the same multiset distribution as
 *      torch.randint, no sort).  With the pool kept in node order (shine_plan_batch on the whole pool, once per frame)
 *      the drawn batch is already ordered: pass idx_out as `perm` to shine_train_step with cfg->sorted_input = 2.
 *      idx_out [n] int32 ascending; (seed, stream_id) select the random stream (use the iteration number as
 *      stream_id).  zero_ptr/zero_bytes: optional 16-B aligned buffer cleared in the same pass (the gradient bucket).
 *      surf_parts (optional, with surf_bits = one bit per pool sample, bit (i & 31) of word i >> 5 set iff weight[i] > 0):
 *      device int64[SHINE_SURF_PARTS]; the launches that write the indices also count the draws (of the slice, for
 *      shine_sample_sorted_slice) whose bit is set — the surface
 *      samples the eikonal term averages over (shine_batch.py:183-185) — as 64 partial counts (overwritten): hand them to
 *      shine_train_step as n_surf with cfg->n_surf_parts = SHINE_SURF_PARTS.  workspace == NULL returns the required
 *      bytes. --------------------------------------------------------------------------------------------------------- */
int shine_sample_sorted(int64_t pool_size, int64_t n, uint64_t seed, uint64_t stream_id, int32_t* idx_out,
                        void* zero_ptr, size_t zero_bytes, const uint32_t* surf_bits, int64_t* surf_parts, void* workspace,
                        size_t* workspace_bytes, void* stream);

/* the same draw for ONE data-parallel rank: only draws [slice_begin, slice_begin + slice_n) of the global sorted batch of
 * n draws are written (idx_out [slice_n]); every rank passes the same (seed, stream id) and its own slice, so the ranks'
 * slices are the contiguous parts of one i.i.d. batch at 1/world of the writing cost.  stream_state: device
 * uint64[4] (graph-replayable, as shine_sample_sorted_dev) or NULL to use stream_id. */
int shine_sample_sorted_slice(int64_t pool_size, int64_t n, int64_t slice_begin, int64_t slice_n, uint64_t seed,
                              uint64_t stream_id, uint64_t* stream_state, int32_t* idx_out, void* zero_ptr,
                              size_t zero_bytes, const uint32_t* surf_bits, int64_t* surf_parts, void* workspace,
                              size_t* workspace_bytes, void* stream);
/* the second half of a graph-replayable draw whose pass 1 rode on the previous shine_train_step (cfg->next_draw): ONE launch
 * (index writing, surface count, ride-along clear); arguments as shine_sample_sorted_slice with a device stream state.  Valid
 * ONLY as the first use of that stream state after such a step, for the same n (the workspace then holds this draw's block
 * sums and the state's shadow word names it); anything else must use shine_sample_sorted_dev / _slice. */
int shine_sample_sorted_finish(int64_t pool_size, int64_t n, int64_t slice_begin, int64_t slice_n, uint64_t seed,
                               uint64_t* stream_state, int32_t* idx_out, void* zero_ptr, size_t zero_bytes,
                               const uint32_t* surf_bits, int64_t* surf_parts, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ---- data-parallel exchange of the rows a step touched (the reference is single-GPU: no counterpart).
 *      flags[l]: uint8 [rows[l]] from shine_mark_touched (OR-reduced over the ranks), rows[l] = the level's row count
 *      WITHOUT the trash row (= the trash row's index).
 *      shine_touched_index: ascending ids of the flagged rows per level -> idx_out[l] (capacity rows[l]) and
 *        counts_dev[l] (device int64); workspace == NULL returns the required bytes.
 *      shine_touched_pack: msg = [the listed rows of level 0, level 1, ... (8 floats each)][the L trash rows]; counts =
 *        the HOST copy of counts_dev (the message size is needed on the host for the collective anyway).
 *      shine_touched_unpack: the reverse; clears flags[l] at the listed rows when flags is given. */
int shine_touched_index(int32_t n_levels, const uint8_t* const* flags, const int64_t* rows, int32_t* const* idx_out,
                        int64_t* counts_dev, void* workspace, size_t* workspace_bytes, void* stream);
int shine_touched_pack(int32_t n_levels, float* const* grads, const int32_t* const* idx, const int64_t* counts,
                       const int64_t* rows, float* msg, void* stream);
int shine_touched_unpack(int32_t n_levels, float* const* grads, const int32_t* const* idx, const int64_t* counts,
                         const int64_t* rows, uint8_t* const* flags, const float* msg, void* stream);

/* ---- the same exchange with ONE collective, no host read and half the bytes: every rank packs only the rows IT touched and
 *      the ranks all-gather the messages (no reference counterpart).  The gradient bucket is one flat float
 *      array: n_rows 8-float rows (the L dense feature-grad tables [rows_l + 1][8] back to back), then — at float offset
 *      tail_off — a dense tail of tail_n floats (the decoder's grads).  flags: one byte per row (the step's touched flags
 *      of all levels back to back).  A message is shine_rows_message_words(cap, tail_n) 4-byte words:
 *      {count, overflow, 0, 0 | ids int32[cap] | values float[cap][8] | tail float[tail_n]}, cap a multiple of 4.
 *      shine_rows_pack: the flagged rows, ascending, MOVE into msg (zeroed in the bucket, flags cleared except keep_rows —
 *        the host array of the <= 8 trash-row ids, whose flags stay set: every miss lands there), and so does the tail.  A
 *        second micro-batch of the same step can therefore accumulate into the same bucket and be packed separately.  More
 *        than cap flagged rows: msg[1] = 1 and the excess rows stay behind (the result is then incomplete — size cap from
 *        a measured step, and check overflow_out).  workspace == NULL returns the required bytes.
 *      shine_rows_unpack_add: msgs = world messages back to back (the all-gather's output, this rank's included): every row
 *        is added into the bucket, one launch per rank in rank order (bit-identical results on every rank), the tails are
 *        summed in rank order and — tail_add == 0 — stored (the first exchange of a step: shine_rows_pack left zeros there) or
 *        — tail_add != 0 — added to what the bucket holds (the later micro-batches of the same step).  tail_n is the tail
 *        length the messages were packed with in either case: it is part of the message stride.  *overflow_out (device
 *        int32, optional) is set to 1 if any rank overflowed. */
int64_t shine_rows_message_words(int64_t cap, int64_t tail_n);
int shine_rows_pack(uint8_t* flags, int64_t n_rows, const int64_t* keep_rows, int32_t n_keep, float* bucket,
                    int64_t tail_off, int64_t tail_n, int64_t cap, int32_t* msg, void* workspace, size_t* workspace_bytes,
                    void* stream);
int shine_rows_unpack_add(const int32_t* msgs, int32_t world, int64_t cap, float* bucket, int64_t tail_off, int64_t tail_n,
                          int32_t tail_add, int32_t* overflow_out, void* stream);

/* ---- Mesher.query_points (utils/mesher.py:33-108): query_feature(coord, faster=True) (model/feature_octree.py:237-244,
 *      :267-286) + Decoder.sdf (model/decoder.py:49-63) for n grid points in one launch.
 *      sdf_out[n] f32 = (negate ? -1 : +1) * sdf   (the mesher negates, mesher.py:69,92), may be NULL;
 *      mask_out[n] u8 = 1 iff hierarchical_indices[check_level] (BOTTOM-UP, 0 = leaf) has all 8 ids >= 0
 *      (mesher.py:78-86, check_level = min(featured_level_num, mc_vis_level) - 1, :47), may be NULL. -------- */
int shine_query_points(const shine_tables* t, const shine_step_config* cfg, const float* coord, int64_t n,
                       const float* const* feats, const int64_t* rows, const float* const* mlp, int32_t check_level,
                       int32_t negate, float* sdf_out, uint8_t* mask_out, void* stream);

/* ---- graph-replayable forms of the two calls whose per-iteration scalars are otherwise baked into a captured HIP
 *      graph: the scalars live in device memory and the kernels advance them, so ONE captured iteration
 *      {draw, shine_train_step, [shine_regularize], Adam} can be replayed for every iteration of a frame.
 *      stream_state: device uint64[4] = {stream id (read, then +1 by the launches), 0, 0, 0} ([1], [2] are the library's: a
 *      block counter and a shadow of the id, through which two launches advance it without atomics);  step_state: device int64[8] =
 *      {optimiser steps taken so far (this launch performs step [0]+1 and stores it), the two bias corrections (floats),
 *      beta1^t and beta2^t as doubles (running products; 0 = derive them with pow once), 4 reserved};  zero_grad: bit 0 =
 *      clear the grads in the same pass, bit 1 = the state was already advanced for this step by shine_train_step
 *      (cfg->adam_state): no preparation launch;  lr_dev: device float[n_tensors]
 *      (step_lr_decay, utils/tools.py:135-155, becomes a small device copy outside the graph). --------------------- */
int shine_sample_sorted_dev(int64_t pool_size, int64_t n, uint64_t seed, uint64_t* stream_state, int32_t* idx_out,
                            void* zero_ptr, size_t zero_bytes, const uint32_t* surf_bits, int64_t* surf_parts, void* workspace,
                            size_t* workspace_bytes, void* stream);
int shine_adam_step_dev(int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, const int64_t* numel, const float* lr_dev, const float* weight_decay,
                        float beta1, float beta2, float eps, int64_t* step_state, int32_t zero_grad,
                        unsigned char* const* row_flags, void* stream);

/* ---- the tail of a training iteration in ONE launch (shine_batch.py:208-210, shine_incre.py:152-181): after a
 *      shine_train_step with cfg->defer_reduce = 1 (same cfg and n; workspace = that call's workspace, n_surf as given there)
 *        - the step's per-workgroup partial sums are added up where they are consumed (decoder grads, trash-row grads) and
 *          loss_parts[4] is written as shine_train_step would have;
 *        - lambda_forget != 0: FeatureOctree.cal_regularization (model/feature_octree.py:246-255) on the rows the step
 *          flagged in touched[s] (flags cleared): *reg_out += the unweighted value (cleared by the step: cfg->zero_f64 =
 *          reg_out), gradient 2 lambda importance (F - F_last) unless grad_on[s] == 0 — as shine_regularize;
 *        - torch's Adam on every tensor, gradients cleared — as shine_adam_step_dev with zero_grad = 1 | 2: step_state was
 *          advanced by the step (cfg->adam_state = step_state), lr_dev[lr_index[i]] is tensor i's learning rate.
 *      Tensors: the L feature tables TOP-DOWN ([rows_l + 1][8]), then the decoder's W1, b1, W2, b2, w3, b3 (or only the
 *      tables when the decoder is frozen).  next_draw (optional): the sorted draw of the NEXT iteration — what
 *      shine_sample_sorted_dev(pool_size, n, seed, stream_state, idx_out, NULL, 0, surf_bits, surf_parts, ...) would launch, for
 *      n < 16 K draws — done by a few extra blocks of this launch (the fused kernel is done with the index buffer by then),
 *      so an iteration at the reference's batch size is {fused step, this}: two launches (shine_next_draw: above).
 *      touched[s] (byte per row incl. the trash row, required with lambda_forget != 0 or active_rows): bit 0 = the row received
 *      gradient from THIS iteration's step (set by the step: its `touched` argument).  active_rows == 0: the flags are cleared
 *      here (as shine_regularize does).  active_rows != 0 — EXACT active-row Adam: the flags are sticky (1 becomes 2 = "touched
 *      earlier") and a row whose flag is 0 is skipped without being read: it has had no gradient since the caller cleared
 *      flags, m and v together (a new optimiser, utils/tools.py:57-83 — every frame in shine_incre.py:107-109), so m = v = g = 0
 *      and torch's Adam leaves it bit for bit unchanged (p -= lr * 0 / (0 + eps)).  Needs weight_decay == 0 on the feature
 *      tables (the reference's groups, utils/tools.py:68-72) and grads that are zero on unflagged rows. */
int shine_finish_iteration(const shine_step_config* cfg, int64_t n, const void* workspace, const int64_t* n_surf,
                           double* loss_parts, const float* const* feats_last, const float* const* importance,
                           unsigned char* const* touched, const int32_t* grad_on, float lambda_forget, double* reg_out,
                           int32_t n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
                           float* const* exp_avg_sq, const int64_t* numel, const float* lr_dev, const int32_t* lr_index,
                           const float* weight_decay, float beta1, float beta2, float eps, const int64_t* step_state,
                           const shine_next_draw* next_draw, int32_t active_rows, void* stream);

/* ---- the training iteration as a HIP graph built and re-bound by the library (shine_batch.py:105-210, shine_incre.py:114-181:
 *      the loop body the drivers run `iters` times per frame).  An iteration = { shine_train_step with cfg->defer_reduce = 1,
 *      shine_finish_iteration }: two kernel launches.  The graph holds `unroll` copies of that pair as kernel nodes;
 *      set_step / set_finish take exactly the arguments of the two entry points (minus the stream) and record the launches
 *      they would make; commit builds and instantiates the graph the first time (and whenever a different kernel
 *      instantiation is needed) and otherwise only rewrites the nodes' kernel parameters in the instantiated graph — so a new
 *      frame of incremental mapping (new feature tables, new optimiser state: model/feature_octree.py:147-160,
 *      shine_incre.py:107-109) costs no stream capture and no instantiation.  launch replays it `replays` times on the
 *      stream (= replays * unroll iterations; the per-iteration scalars — sampler stream id, Adam step count — live in device
 *      memory and are advanced by the kernels).  The argument arrays are read inside set_*; the DEVICE buffers they name
 *      must stay alive while the graph is launched.  For small batches (< 32768 points) the graph can keep the decoder's MFMA
 *      operand image in a buffer the CALLER hands it (set_operand_image: shine_iter_graph_operand_image_floats() floats, 16-byte
 *      aligned, alive as long as the graph; call it before set_step; NULL or never: no image, the step builds its operands per
 *      workgroup — the library allocates no device memory for a graph): launch rebuilds it first from the decoder tensors as they
 *      are then, the tail nodes keep it current, the step nodes copy it instead of building it per workgroup.  commit waits (on the host) for the graph's own last replay before it touches
 *      the instantiated graph, not for other work queued since.  stats: commits so far, and how many of them built the graph. */
typedef struct shine_iter_graph shine_iter_graph;
int shine_iter_graph_create(int32_t unroll, shine_iter_graph** out);
int shine_iter_graph_destroy(shine_iter_graph* g);
int shine_iter_graph_operand_image_floats(void);
int shine_iter_graph_set_operand_image(shine_iter_graph* g, float* image);
int shine_iter_graph_set_step(shine_iter_graph* g, const shine_tables* t, const shine_step_config* cfg, const float* coord,
                              const float* sdf_label, const float* weight, const int32_t* perm, const int32_t* slots,
                              const int64_t* n_surf, int64_t n, const float* const* feats, const int64_t* rows,
                              const float* const* mlp, float* pred_out, float* grad_x_out, float* const* grad_feats,
                              float* const* grad_mlp, double* loss_parts, unsigned char* const* touched, void* workspace,
                              size_t workspace_bytes);
int shine_iter_graph_set_finish(shine_iter_graph* g, const shine_step_config* cfg, int64_t n, const void* workspace,
                                const int64_t* n_surf, double* loss_parts, const float* const* feats_last,
                                const float* const* importance, unsigned char* const* touched, const int32_t* grad_on,
                                float lambda_forget, double* reg_out, int32_t n_tensors, float* const* params,
                                float* const* grads, float* const* exp_avg, float* const* exp_avg_sq, const int64_t* numel,
                                const float* lr_dev, const int32_t* lr_index, const float* weight_decay, float beta1,
                                float beta2, float eps, const int64_t* step_state, const shine_next_draw* next_draw,
                                int32_t active_rows);
int shine_iter_graph_commit(shine_iter_graph* g);
int shine_iter_graph_launch(shine_iter_graph* g, int32_t replays, void* stream);
int shine_iter_graph_stats(const shine_iter_graph* g, int64_t* commits, int64_t* builds);

/* ---- measurement aid (tools/ab_build.py AB_PROF): per-wave phase cycle counters of the fused kernel.  buffer = device
 *      int64 [waves][8] (setup, query, decoder forward, loss+backward, scatter, weight grads, flush, block wait) that
 *      the next 4-level shine_train_step launches fill through s_memtime stamps; NULL switches it off again.
 *      The stamped build is a separate template instantiation: the product kernel carries no profiling code. -------- */
void shine_debug_set_profile_buffer(int64_t* buffer);

/* ---- device self-test: D[16,16] = A[16,4] . B[4,16] through ONE v_mfma_f32_16x16x4_f32, written back with the
 *      accumulator lane map the fused kernel relies on (pins the MFMA operand layouts on the hardware). --- */
int shine_selftest_mfma16(const float* a, const float* b, float* d, void* stream);
/* cross-lane exchanges of the lane = (point, level) kernel (shine_step_v3.hip), 64 lanes each: through
 *  v_permlane32_swap  o32[l] = l < 32 ? x[l] + x[l + 32] : y[l] + y[l - 32];
 *  v_permlane16_swap  o16[l] = (l & 16) ? y[l] + y[l - 16] : x[l] + x[l + 16]                                  */
int shine_selftest_permlane(const float* x, const float* y, float* o32, float* o16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SHINE_HIP_H_ */
