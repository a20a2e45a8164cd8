// shine_step_body.hpp — the body of the fused SHINE training step (one workgroup's share of a batch) as a device function,
// so that two kernels can run it: k_step_v3 (shine_step_v3.hip: one launch = one step) and k_step_sweep (shine_sweep.hip: the
// SLICED build — up to 64 independent chunk steps of the importance sweep in one launch).  The design notes are in
// shine_step_v3.hip's header.
#pragma once
#include "shine_tile16.hpp"

namespace shine {

#ifndef SHINE_V3_BIG
#define SHINE_V3_BIG 8
#endif
constexpr int V3_BIG = SHINE_V3_BIG;           // waves per workgroup of the full-chip launch (V3_BIG / 4 per SIMD)
#ifndef SHINE_V3_GB
#define SHINE_V3_GB 4
#endif
constexpr int V3_GB = SHINE_V3_GB;             // corner rows gathered per batch (register budget: 8 floats each)
#ifndef SHINE_V3_ABLATE  // measurement builds only (tools/mk_variant.py -DSHINE_V3_ABLATE=bits; the product is 0): 1 no feature-grad
#define SHINE_V3_ABLATE 0  // atomics, 8 no row gathers (zeros instead) — how much of the kernel each holds exclusively (far_ablate.py);
                           // 16: s_waitcnt vmcnt(0) behind every tile's scatter
#endif
#ifndef SHINE_V3_PROFBUILD  // 1: also instantiate the kernels with per-wave phase cycle counters (AB_PROF of tools/ab_build.py)
#define SHINE_V3_PROFBUILD 0
#endif

// EXT: the backward half of Tier A's fused node (autograd_ops.FusedInterpSdf): d loss / d pred comes from autograd
// (a.ext_delta) instead of the kernel's own BCE — query, decoder forward, decoder backward, weight grads and scatter are the
// same code.  EXT + EIK: autograd also hands over d loss / d g for g = d pred / d coord (a.ext_q, the gradient that reaches
// get_gradient's output, utils/tools.py:175-185): the closed-form eikonal chain then backpropagates THAT instead of the
// kernel's own (1 - |g|)^2 term — whatever the driver built on g (shine_batch.py:182-185 or another loss).
// MARK: the touched-row flags (for shine_regularize) are set by the scatter at the run start of every hit node instead of by a
// k_mark_touched launch in front of the step — a build of its own, because the flag code costs the kernel ~10 % at 2^18
// points even when there are no flags to set (profiles/r03_ab_experiments.txt block 3), and pays at the incremental
// configuration's 4096 points, where the extra launch is half the step (ncd-incre 172 -> 179 frames/s).
// SLICED: the launch holds many independent steps (shine_sweep.hip); `sl` names this workgroup's one — sample indices, size,
// normaliser, private gradient tables and flags — and bid / nbid count inside the slice.
// AHEAD (every full-chip build, WAVES == V3_BIG): a tile's dependent chain  hash slot -> 8 corner ids -> 8 rows  is run one
// tile ahead — the slot arrives two tiles ahead and the corner ids of tile t + 1 are requested under the work of tile t, so the
// row gathers of a tile start from registers.  Round 5, same-box A/B: 71.4 -> 69.6 us (2^18 x L4 BCE), 232 -> 224 us (2^20 x L3
// eikonal), 502 -> 499 us on the 324 MB map (profiles/r05_ab_experiments.txt block 2; round 3 had measured the opposite sign
// on its kernel).  TOUCHING the rows of tile t + 1 into the L2 on top of that — one dword load per row into a register nobody
// reads — makes the big map SLOWER (540 us with the feature rows, 632 us with the gradient rows too: block 1): that regime is
// not waiting for memory, it is short of atomic throughput (below); tools/experiments/r05_far_touch.patch keeps the code.
//
// FAR: the build for feature tables beyond the Infinity Cache (the launch picks it by table size, shine_step_v3.hip).  What
// bounds the step there is the L2's fp32 atomic rate: the run-merged scatter of a 2^20-point batch on the 324 MB map is 1.28 M
// node runs = 82 M lane atomics, which the memory system retires in 408 us with NOTHING else running
// (tools/ubench/random_rows.py: ~200 per ns, the same rate as on a cache-resident map) inside a 500 us kernel — the gathers
// of the same batch take 106 us.  Node runs merge little on such a map (about one sample per leaf node), but neighbouring nodes
// share corner rows: the FAR build keeps, per wave and level, a 4 x 4 x 4 lattice of corner rows in LDS (slot = the corner's
// voxel coordinates mod 4, tag = its row id), adds a closed run's eight row sums there, and issues the global atomic only when
// a slot is taken over by another row or the wave is done: 2.0-2.2 x fewer atomics on the synthetic maps
// (profiles/r05_corner_merge_sim.txt).  The eight corners of one node always fall into eight different slots, so the eight
// lane groups of the scatter never meet in one.
// FAR: a node run's sum for ONE corner row goes into the wave's lattice of that level.  Called by all 64 lanes of the scatter
// role at once, lane = (corner, feature); `packed` = the corner's row id | its lattice slot << 25.  lattice_peek — when the run
// STARTS (or is carried into a tile) — reads the slot's tag and value; lattice_close — when the run ends — only writes: the slot
// held the row already: value + sum; it held another row (or nothing): that row's value leaves as one atomic per feature and the
// slot is taken over.  So the serial walk never waits for an LDS round trip at a run's end (the first version — peek and close in
// one — cost the cache-resident kitti map 298 us against 225: profiles/r05_ab_experiments.txt block 4).  The eight corners of a
// node sit in eight different slots, so no two lane groups meet in one; nothing else writes the lattice between a run's peek
// and its close (closes are the only writes, and a wave closes one run per level at a time).
__device__ __forceinline__ void grad_atomic(float* p, float v) {
  if (!(SHINE_V3_ABLATE & 1)) atomic_add_f32(p, v);
}
__device__ __forceinline__ void lattice_peek(const float* lat, int packed, int sq, int& T, float& V) {
  const int slot = (packed >> V3_ROW_BITS) & (V3_CSLOTS - 1);
  T = reinterpret_cast<const int*>(lat)[slot];
  V = lat[V3_CSLOTS + slot * F + sq];
}
__device__ __forceinline__ void lattice_close(float* lat, float* gbase, int packed, int sq, float v, int T, float V) {
  const int row = packed & ((1 << V3_ROW_BITS) - 1), slot = (packed >> V3_ROW_BITS) & (V3_CSLOTS - 1);
  const bool same = T == row;
  lat[V3_CSLOTS + slot * F + sq] = same ? V + v : v;
  reinterpret_cast<int*>(lat)[slot] = row;  // (the group's eight lanes write the same word)
  if (!same && T >= 0) grad_atomic(gbase + ((unsigned int)T << 3) + sq, V);
}

// RECORD pools (a.pool_mode == 2, cfg->sorted_input 3): the pool is ONE 32-byte record per sample instead of four arrays read at
// the same sparse sorted positions — {x, y, z, label | weight, slot[0 .. L-1], pad} for L <= 3, {x, y, z, label | slot[0 .. 3]} for
// L = 4 (the weight, read by the eikonal / weighted builds only, stays a separate array there) — so a drawn sample costs one
// cache line, not four (2^20 samples x 4 lines x 128 B were 537 of the kitti step's 762 MB of HBM traffic).
// a.coord is the record base.  rec_words: the record as dwords; the hash slot of level g sits at dword REC_SLOT0 + g.
template <int L>
struct RecLayout {
  static constexpr int SLOT0 = L == 4 ? 4 : 5;  // first slot dword
  static constexpr bool WEIGHT_INSIDE = L < 4;  // dword 4
};
__device__ __forceinline__ const int* rec_words(const V1Args& a, long long sample) {
  return reinterpret_cast<const int*>(a.coord) + 8 * sample;
}

template <int L, int WAVES, bool EIK, bool PROF, bool EXT = false, bool MARK = false, bool SLICED = false, bool FAR = false>
__device__ __forceinline__ void step_body(const V1Args& a, StepShared<WAVES, FAR>& sm, const int bid, const int nbid,
                                          const StepSlice* sl = nullptr) {
// (spelled as expressions at every use, not as locals: the unsliced builds must stay the instruction streams the committed
// counter files were measured on — tools/kernel_hash.py)
#define SL_N (SLICED ? sl->n : a.n)
#define SL_PERM (SLICED ? sl->perm : a.perm)
#define SL_TOUCHED(s) (SLICED ? sl->touched[s] : a.touched[s])
  constexpr int NT = WAVES * 64;
  float* const s_opA = sm.opA;
  float* const s_bias = sm.bias;
  double* const s_loss = sm.loss;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pt = lane & 15, g = lane >> 4;
  const bool poly = a.poly != 0;
  // Small-batch builds (4 waves) may be handed the decoder's operand image ready-made (a.op_image: kept current by the iteration
  // tail, shine_graph.hip): three 16-byte loads per thread instead of eleven permuted ones with their address arithmetic — at
  // one tile per wave the prologue's instructions are a fifth of the kernel (profiles/r04_ab_experiments.txt blocks 12, 14).
  // They are the FIRST loads of the kernel: their addresses need nothing but the thread id, and the tile split below (two
  // 64-bit divisions) runs under their round trip.
  float4 img[3];
  const bool use_img = WAVES == 4 && a.op_image != nullptr;
  if (WAVES == 4 && use_img) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      img[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tid + k * NT < V3_IMAGE_FLOATS / 4) img[k] = reinterpret_cast<const float4*>(a.op_image)[tid + k * NT];
    }
  }
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tk = PROF ? clk() : 0;
#define SHINE_STAMP(k)            \
  if (PROF) {                     \
    long long now__ = clk();      \
    pc[k] += now__ - tk;          \
    tk = now__;                   \
  }

  // Tiles: workgroup b owns tiles [b T / B, (b + 1) T / B) of the ordered stream, its waves contiguous parts of that.
  // The two waves of a SIMD are not served equally (issue arbitration prefers the older one: the per-wave cycle counters
  // show waves 0-3 of an 8-wave workgroup finishing ~20 % before waves 4-7 on equal shares), so the first half of the
  // waves takes OLD_SHARE / 256 of the workgroup's tiles where the wave's run is long enough for it to matter (A/B at
  // 2^20 points x 3 levels with the eikonal term, 32 tiles per wave: 140 -> -3.6 %; at 2^18 x 4 BCE, 8 tiles per wave,
  // an even split is best).
  constexpr int OLD_SHARE = EIK ? 140 : 128;
  const long long wave_g = (long long)bid * WAVES + wv;
  long long begin, end_t;
  {
    const long long t0 = ((long long)bid * a.tiles) / nbid, t1 = ((long long)(bid + 1) * a.tiles) / nbid;
    const long long nt = t1 - t0;
    constexpr int HW = WAVES / 2;
    const long long cut = (WAVES >= 8 && nt >= 4 * WAVES) ? (nt * OLD_SHARE) >> 8 : nt / 2;  // tiles of waves [0, HW)
    long long lo = wv < HW ? (wv * cut) / HW : cut + ((wv - HW) * (nt - cut)) / HW;
    long long hi = wv < HW ? ((wv + 1) * cut) / HW : cut + ((wv - HW + 1) * (nt - cut)) / HW;
    if (a.ablate & 64) {  // deterministic accumulation (tests): ONE wave of the one-workgroup launch walks the whole stream,
      lo = wv == 0 ? 0 : nt;  // so every feature-grad atomic is issued — and applied — in stream order
      hi = nt;
    }
    begin = V3_TP * (t0 + lo);
    end_t = V3_TP * (t0 + hi);
  }
  const long long end = end_t < SL_N ? end_t : SL_N;
  const long long second = begin + V3_TP;

  // ---- the first loads of the launch, in dependency order so that their round trips overlap (a wave's fixed cost was
  // 12 k cycles of setup in front of ~6 k cycles per tile: profiles/r03_ab_experiments.txt block 7): (1) the sample indices
  // of the first two tiles, (2) the decoder weights for the operand image, (3) — once (1) is back — the first tile's point
  // data and hash slots; only then are the weights waited for and stored to LDS.
  // software prefetch of the {perm -> coord, label, slot} chain, index two tiles ahead
  const int lvl_on_i = g < L;
  long long np = 0;
  float nx0 = 0.f, nx1 = 0.f, nx2 = 0.f, nlabel = 0.f, nweight = 0.f;
  int nslot = -1;
  bool nvalid = begin + pt < end;
  int np2 = 0, np1 = 0;
  if (SL_PERM && second + pt < end) np2 = SL_PERM[second + pt];
  if (SL_PERM && nvalid) np1 = SL_PERM[begin + pt];
  constexpr bool AHEAD = WAVES == V3_BIG && !SLICED;
  int np3 = 0;  // AHEAD: the sample index of the tile after that (its slot is loaded two tiles ahead)
  if (AHEAD && SL_PERM && second + V3_TP + pt < end) np3 = SL_PERM[second + V3_TP + pt];

  // ---- per-workgroup setup: A operands in 16x16x4 lane order (lane l: row i = l & 15, k = l >> 4), biases.
  // Branch-free source select and a fully unrolled loop: the (up to 11) loads of a thread are all in flight together.
  OperandRegs<NT> opr;
  if (!(WAVES == 4 && use_img)) decoder_operands_issue<NT>(a, opr, tid);

  float inv_nsurf = 0.f;
  if (EIK) {  // the batch's surface count: one number, or the sampler's per-block parts (cfg->n_surf_parts) added up here
    long long ns = 0;
    if (a.n_surf) {
      if (a.n_surf_parts <= 1) {
        ns = *a.n_surf;
      } else {
        int c = lane < a.n_surf_parts ? (int)a.n_surf[lane] : 0;  // (<= 64 parts: one load per lane)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        ns = c;
      }
    }
    inv_nsurf = ns > 0 ? 1.0f / (float)ns : 0.f;
    // the count this step used, for the launch that finalises the loss (k_reduce_partials / k_finish): the parts themselves
    // may be overwritten by then — the next draw, riding on that very launch, counts ITS surface samples into them
    if (bid == 0 && tid == 0) reinterpret_cast<double*>(a.partials + PART_LOSS)[3] = (double)ns;
  }

  // pool batches of the one-step launches are RECORD pools (prepare_step_v3 refuses the array form); the sliced sweep keeps the
  // arrays (its frame pools are small and planned once)
  const bool rec = !SLICED && a.pool_mode != 0;  // (wave-uniform: scalar branches)
  const bool soa_pool = SLICED && a.pool_mode != 0;
  const bool need_w = (EIK && !EXT) || a.weighted;
  if (nvalid) {
    np = SL_PERM ? (long long)np1 : begin + pt;
    if (rec) {
      const int* r = rec_words(a, np);
      if (lvl_on_i) nslot = r[RecLayout<L>::SLOT0 + g];
      const float4 c = *reinterpret_cast<const float4*>(r);
      nx0 = c.x, nx1 = c.y, nx2 = c.z, nlabel = c.w;
      if (need_w) nweight = RecLayout<L>::WEIGHT_INSIDE ? __int_as_float(r[4]) : a.weight[np];
    } else {
      const long long si = soa_pool ? np : begin + pt;
      if (lvl_on_i) nslot = __builtin_nontemporal_load(a.slots + si * L + g);
      nx0 = a.coord[3 * np];
      nx1 = a.coord[3 * np + 1];
      nx2 = a.coord[3 * np + 2];
      nlabel = a.label[np];
      if (need_w) nweight = a.weight[np];
    }
  }
  int fslot = -1;  // AHEAD: the hash slot of the tile after the one n* describes
  if (AHEAD && lvl_on_i && second + pt < end)
    fslot = rec ? rec_words(a, np2)[RecLayout<L>::SLOT0 + g]
                : __builtin_nontemporal_load(a.slots + (soa_pool ? (long long)np2 : second + pt) * L + g);

  if (WAVES == 4 && use_img) {  // (opA and bias are adjacent in StepShared: the image is their concatenation)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      if (tid + k * NT < V3_IMAGE_FLOATS / 4) reinterpret_cast<float4*>(s_opA)[tid + k * NT] = img[k];
  } else {
    decoder_operands_store<NT>(opr, s_opA, s_bias, tid);
  }
  if (tid == 0) s_loss[0] = s_loss[1] = s_loss[2] = s_loss[3] = 0.0;
  __syncthreads();

  // FAR: this wave's corner lattice (tags -1 = empty); only this wave touches it
  float* const cbase = sm.cache[FAR ? wv : 0];
  if (FAR) {
#pragma unroll
    for (int s = 0; s < L; ++s) reinterpret_cast<int*>(cbase)[s * V3_CLEVEL + lane] = -1;
  }
  float* U = sm.wave[wv];
  int* U_ids = reinterpret_cast<int*>(U);  // [LCAP][8][16]
  float* U_w = U + V3_IDS;                 // [LCAP][8][16]
  float* R2 = U + V3_IDS + V3_W;

  // Per-lane LDS base addresses: every staging access below is one of these + a compile-time offset (DS instructions
  // carry a 16-bit immediate).  They are re-derived from an opaque lane value at the top of every tile, which keeps
  // LLVM's loop-invariant code motion from parking pre-added address variants in VGPRs across the whole loop.
  int lane_o = lane;
  const float b3 = s_bias[96];
  const float inv_sigma = 1.0f / a.sigma;
  const float4* sb4 = reinterpret_cast<const float4*>(s_bias);

  // this lane's level (query role): table pointers and resolution of level g (lanes of a level the tree does not have
  // borrow the leaf level's pointers and never hit)
  const bool lvl_on = g < L;
  const int gs = lvl_on ? g : L - 1;
  const float* lv_feat = a.lv[0].feat;
  const int4* lv_vals = a.lv[0].vals;
  float lv_res = a.lv[0].res;
#pragma unroll
  for (int s = 1; s < L; ++s)
    if (gs == s) {
      lv_feat = a.lv[s].feat;
      lv_vals = a.lv[s].vals;
      lv_res = a.lv[s].res;
    }

  int4 fia = make_int4(0, 0, 0, 0), fib = fia;  // AHEAD: the corner ids of the tile n* describes, loaded one tile ahead
  if (AHEAD) {
    const unsigned int s0 = nvalid && nslot >= 0 ? (unsigned int)nslot : 0u;
    fia = lv_vals[2u * s0];
    fib = lv_vals[2u * s0 + 1u];
  }

  f32x4 accW2[2][2], accW1[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    accW1[m] = zero4();
#pragma unroll
    for (int n = 0; n < 2; ++n) accW2[m][n] = zero4();
  }
  f32x2 dw3c[4];  // channel pairs (2 j, 2 j + 1) of this lane's eight (packed fp32: half the FMAs)
#pragma unroll
  for (int j = 0; j < 4; ++j) dw3c[j] = splat2(0.f);
  float db2acc[2] = {0.f, 0.f};  // BCE build: db2 rides on the transposed operands of the dW2 pass
  f32x2 db2c[EIK ? 4 : 1], db1c[EIK ? 4 : 1];  // eikonal build: sum_p delta_p v2 / v1 for this lane's channels
#pragma unroll
  for (int j = 0; j < (EIK ? 4 : 1); ++j) db2c[j] = db1c[j] = splat2(0.f);
  float eik_acc = 0.f;
  float db3 = 0.f;
  float loss_acc = 0.f;  // per-lane sum over this wave's <= a few dozen tiles; widened to double at the flush
  int cnt_acc = 0;
  int run_id[LCAP], run_hit[LCAP];
  float run_acc[LCAP];
  float trash_sum = 0.f;
#pragma unroll
  for (int s = 0; s < LCAP; ++s) {
    run_id[s] = -1;
    run_acc[s] = 0.f;
    run_hit[s] = 0;
  }
  int last_slot = -2;  // this lane's level: the node of the previous tile's last point (carries runs across tiles)
  const int sc = lane >> 3, sq = lane & 7;  // scatter role: corner, feature
  SHINE_STAMP(0)  // setup

  for (long long base = begin; base < end; base += V3_TP) {
    asm volatile("" : "+v"(lane_o));  // opaque per tile (see above)
    const int o_pt = lane_o & 15, o_g = lane_o >> 4;
    int* const st_ids = U_ids + (8 * o_g) * V3_WP + o_pt;   // staging writes (level o_g): + c * V3_WP
    float* const st_w = U_w + (8 * o_g) * V3_WP + o_pt;
    const int* const sc_ids = U_ids + (lane_o >> 3) * V3_WP;  // scatter reads: + s * 8 * V3_WP + point
    const float* const sc_w = U_w + (lane_o >> 3) * V3_WP;
    const float* const sc_df = R2 + V3_DF + (lane_o & 7) * V3_DFP;
    float* const t_wr = R2 + (4 * o_g) * V3_TT + o_pt;       // transpose writes: + (16 m + r) * V3_TT [+ 32 * V3_TT]
    const float* const t_rd = R2 + o_pt * V3_TT + 4 * o_g;   // operand reads (i16 = lane & 15, kk = lane >> 4)
    float* const f_wr = R2 + (2 * o_g) * V3_TT + o_pt;       // [feature 2g (+1)][pt] rows (V3_DFP == V3_TT)
    const float* const opa = s_opA + lane_o;
    const bool valid = nvalid;
    const long long p = np;
    const long long po = a.pool_mode ? base + pt : p;  // where this point's outputs go
    const float x0 = nx0, x1 = nx1, x2 = nx2, label = nlabel, wgt = nweight;
    // ================================================================ phase 1: query (this lane: level g of point pt)
    const int slot = valid ? nslot : -1;
    const bool hit = slot >= 0;
    const unsigned int validmask = (unsigned int)__ballot(valid) & 0xFFFFu;
    // node-run boundaries of the ordered stream, all levels at once: bit 16 g + pt of one 64-bit ballot
    const int prev = row_prev(slot, last_slot);
    const bool chg = valid && lvl_on && slot != prev;
    const unsigned long long chg64 = __ballot(chg);
    const unsigned long long hit64 = __ballot(hit);
    last_slot = row_last(slot);
    // smooth-step weights of this level, in the reference's association (model/feature_octree.py:186-193)
    float w[8];
    Axis X = axis_weight_rt(poly, x0, lv_res), Y = axis_weight_rt(poly, x1, lv_res), Z = axis_weight_rt(poly, x2, lv_res);
    if (EIK && !hit) X.dt = Y.dt = Z.dt = 0.f;  // a miss: every d w_c / d x carries exactly one of these factors
    corner_weights(X.t, Y.t, Z.t, w);
    if (!hit) {
#pragma unroll
      for (int c = 0; c < 8; ++c) w[c] = 0.f;  // padding lanes and misses contribute nothing to f or to the scatter
    }
    if (!EIK) {  // staging for the scatter: [level][corner][point] (the eikonal build stages its weights after the decoder)
#pragma unroll
      for (int c = 0; c < 8; ++c) st_w[c * V3_WP] = w[c];
    }
    // Packed fp32 (v_pk_fma_f32, two FMAs per lane and instruction): the sums over a row's features are kept as feature PAIRS
    // with the corner's weight broadcast to both halves (op_sel) — per feature the same fmaf chain over the corners, in the same
    // order (model/feature_octree.py:231), half the instructions.
    f32x2 pf[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pf[q] = splat2(0.f);
    f32x2 Ag[EIK ? 4 : 1][3];  // eikonal build: this level's part of d f_q / d x_e, features (2 j, 2 j + 1)
#pragma unroll
    for (int q = 0; q < (EIK ? 4 : 1); ++q) Ag[q][0] = Ag[q][1] = Ag[q][2] = splat2(0.f);
    {  // every lane gathers the 8 ids and the 8 x 32-B rows of its own (point, level)
      const unsigned int sl = hit ? (unsigned int)slot : 0u;
      int4 ia, ib;  // the eight corner ids: two 16-B loads (AHEAD: requested one tile ago)
      if (AHEAD) {
        ia = fia, ib = fib;
      } else {
        ia = lv_vals[2u * sl], ib = lv_vals[2u * sl + 1u];
      }
      const int ids[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
      // staging for the scatter: a miss stages -1 (trash row), never the speculative ids
      const int mneg = hit ? 0 : -1;
      if (FAR) {  // each id carries the lattice slot of its corner: the node's voxel coordinates + the corner's offset, mod 4
        const int vx = (int)__fmul_rn(lv_res, __fadd_rn(__fmul_rn(x0, 0.5f), 0.5f));
        const int vy = (int)__fmul_rn(lv_res, __fadd_rn(__fmul_rn(x1, 0.5f), 0.5f));
        const int vz = (int)__fmul_rn(lv_res, __fadd_rn(__fmul_rn(x2, 0.5f), 0.5f));
        const int sx[2] = {(vx & 3) << (V3_ROW_BITS + 4), ((vx + 1) & 3) << (V3_ROW_BITS + 4)};
        const int sy[2] = {(vy & 3) << (V3_ROW_BITS + 2), ((vy + 1) & 3) << (V3_ROW_BITS + 2)};
        const int sz[2] = {(vz & 3) << V3_ROW_BITS, ((vz + 1) & 3) << V3_ROW_BITS};
#pragma unroll
        for (int c = 0; c < 8; ++c) st_ids[c * V3_WP] = (ids[c] | sx[(c >> 2) & 1] | sy[(c >> 1) & 1] | sz[c & 1]) | mneg;
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) st_ids[c * V3_WP] = ids[c] | mneg;
      }
      // V3_GB corners (2 V3_GB 16-B loads) in flight at a time
#pragma unroll
      for (int cb = 0; cb < 8; cb += V3_GB) {
        float4 r0[V3_GB], r1[V3_GB];
#pragma unroll
        for (int c = 0; c < V3_GB; ++c) {  // a miss reads row 0 with weight 0 (no branches)
          const float* row = lv_feat + (size_t)(hit ? (unsigned int)ids[cb + c] : 0u) * F;
          if (SHINE_V3_ABLATE & 8) {
            r0[c] = r1[c] = make_float4(0.01f * (float)ids[cb + c], 0.f, 0.f, 0.f);  // (keeps the ids live, no row traffic)
            continue;
          }
          r0[c] = *reinterpret_cast<const float4*>(row);
          r1[c] = *reinterpret_cast<const float4*>(row + 4);
        }
#pragma unroll
        for (int cp = 0; cp < V3_GB; cp += 2) {  // corners (cb + cp, cb + cp + 1): the same (cx, cy), cz = 0 / 1
          f32x2 dw[3];
          if (EIK) corner_dw_pair(X, Y, Z, (cb + cp) >> 1, dw);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int c = cp + h;
            const f32x2 rr[4] = {{r0[c].x, r0[c].y}, {r0[c].z, r0[c].w}, {r1[c].x, r1[c].y}, {r1[c].z, r1[c].w}};
            const f32x2 wc = splat2(w[cb + c]);
#pragma unroll
            for (int j = 0; j < 4; ++j) pf[j] = pk_fma(wc, rr[j], pf[j]);
            if (EIK) {
#pragma unroll
              for (int e = 0; e < 3; ++e) {
                const f32x2 de = splat2(dw[e][h]);  // (zero for a miss: dt = 0 above)
#pragma unroll
                for (int j = 0; j < 4; ++j) Ag[j][e] = pk_fma(de, rr[j], Ag[j][e]);
              }
            }
          }
        }
        if (V3_GB < 8) __builtin_amdgcn_sched_barrier(0);
      }
    }
    // prefetch of tile t+1's point data, issued after every gather of this tile (vmcnt counts in order)
    {
      const long long ni = base + V3_TP + pt, ni2 = ni + V3_TP, ni3 = ni2 + V3_TP;
      nvalid = ni < end;
      np = 0;
      nx0 = nx1 = nx2 = nlabel = nweight = 0.f;
      nslot = -1;
      if (AHEAD) {  // the corner ids of tile t+1 (its slot came in a tile ago), then the slot of tile t+2
        nslot = nvalid ? fslot : -1;
        const unsigned int fs = nslot >= 0 ? (unsigned int)nslot : 0u;
        fia = lv_vals[2u * fs];
        fib = lv_vals[2u * fs + 1u];
        fslot = -1;
        if (lvl_on && ni2 < end)
          fslot = rec ? rec_words(a, np3)[RecLayout<L>::SLOT0 + g]  // (brings the record's line in: tile t+2's point data hit it)
                      : __builtin_nontemporal_load(a.slots + (soa_pool ? (long long)np3 : ni2) * L + g);
      }
      if (nvalid) {
        np = SL_PERM ? (long long)np2 : ni;
        if (rec) {
          const int* r = rec_words(a, np);
          if (!AHEAD && lvl_on) nslot = r[RecLayout<L>::SLOT0 + g];
          const f32x4 c = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(r));
          nx0 = c[0], nx1 = c[1], nx2 = c[2], nlabel = c[3];
          if (need_w)
            nweight = RecLayout<L>::WEIGHT_INSIDE ? __int_as_float(__builtin_nontemporal_load(r + 4))
                                                  : __builtin_nontemporal_load(a.weight + np);
        } else {
          const long long si = soa_pool ? np : ni;
          if (!AHEAD && lvl_on) nslot = __builtin_nontemporal_load(a.slots + si * L + g);
          nx0 = __builtin_nontemporal_load(a.coord + 3 * np);
          nx1 = __builtin_nontemporal_load(a.coord + 3 * np + 1);
          nx2 = __builtin_nontemporal_load(a.coord + 3 * np + 2);
          nlabel = __builtin_nontemporal_load(a.label + np);
          if (need_w) nweight = __builtin_nontemporal_load(a.weight + np);
        }
      }
      if (AHEAD) {
        np2 = np3;
        np3 = 0;
        if (SL_PERM && ni3 < end) np3 = __builtin_nontemporal_load(SL_PERM + ni3);
      } else if (SL_PERM && ni2 < end) {
        np2 = __builtin_nontemporal_load(SL_PERM + ni2);
      }
    }
    // reduce-scatter of the per-level sums over the point's four lanes: lane g ends with features (2g, 2g+1)
    float f2[2];
    {
      float h4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) h4[q] = xsum32(pf[q >> 1][q & 1], pf[2 + (q >> 1)][q & 1]);  // g < 2: features q, g >= 2: features 4 + q
#pragma unroll
      for (int t = 0; t < 2; ++t) f2[t] = xsum16(h4[t], h4[2 + t]);  // even g: t, odd g: 2 + t
    }
    float A2[2][3];  // eikonal build: d f_{2g + t} / d x_e of this lane's point (all levels summed)
#pragma unroll
    for (int t = 0; t < 2; ++t) A2[t][0] = A2[t][1] = A2[t][2] = 0.f;
    if (EIK) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        float h4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) h4[q] = xsum32(Ag[q >> 1][e][q & 1], Ag[2 + (q >> 1)][e][q & 1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) A2[t][e] = xsum16(h4[t], h4[2 + t]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // phase boundary: no operand of the next phase is fetched early
    SHINE_STAMP(1)  // query

    // ================================================================ phase 2: decoder forward (MFMA chain)
    f32x4 c1[2], c2[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float4 v1b = sb4[4 * m + g], v2b = sb4[8 + 4 * m + g];
      c1[m][0] = v1b.x, c1[m][1] = v1b.y, c1[m][2] = v1b.z, c1[m][3] = v1b.w;
      c2[m][0] = v2b.x, c2[m][1] = v2b.y, c2[m][2] = v2b.z, c2[m][3] = v2b.w;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m) c1[m] = mfma16(opa[V3_OPA1 + (2 * m + t) * 64], f2[t], c1[m]);
    float h1[8], h2[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) h1[r] = fmaxf(c1[r >> 2][r & 3], 0.f);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int m = 0; m < 2; ++m) c2[m] = mfma16(opa[V3_OPA2 + (8 * m + ks) * 64], h1[ks], c2[m]);
    float yp = 0.f;
    float w3r[8];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const float4 v = sb4[16 + 4 * m + g];
      w3r[4 * m] = v.x, w3r[4 * m + 1] = v.y, w3r[4 * m + 2] = v.z, w3r[4 * m + 3] = v.w;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      h2[r] = fmaxf(c2[r >> 2][r & 3], 0.f);
      yp = fmaf(w3r[r], h2[r], yp);
    }
    yp += __shfl_xor(yp, 16, 64);
    const float y = yp + __shfl_xor(yp, 32, 64) + b3;
    if (!EXT && valid && g == 0 && a.pred) __builtin_nontemporal_store(y, a.pred + po);
    __builtin_amdgcn_sched_barrier(0);  // phase boundary: no operand of the next phase is fetched early
    SHINE_STAMP(2)  // decoder forward

    // ================================================================ phase 3: loss
    // BCEWithLogits(y, z) = max(y, 0) - y z + log1p(e), e = exp(-|y|); sigmoid(y) = 1 / (1 + e) or e / (1 + e) shares e.
    // Hardware transcendentals (v_exp_f32, v_rcp_f32, v_log_f32: ~1 ulp) — errors ~1e-7, the contract is 1e-4.
    float delta = 0.f;
    if (EXT) {
      if (valid) delta = a.ext_delta[po];
    } else {
      const float zt = fast_sigmoid(label * inv_sigma);
      const float e = __builtin_amdgcn_exp2f(-1.44269504088896f * fabsf(y));
      const float r = __builtin_amdgcn_rcpf(1.0f + e);
      const float sg = y >= 0.f ? r : e * r;
      if (valid) {
        const float lw = a.weighted ? fabsf(wgt) : 1.0f;  // BCEWithLogitsLoss(weight=|weight|), utils/loss.py:18-19
        if (g == 0) {
          loss_acc += lw * (fmaxf(y, 0.f) - y * zt + 0.693147180559945f * __builtin_amdgcn_logf(1.0f + e));
          cnt_acc += 1;
        }
        delta = lw * (sg - zt) * (SLICED ? sl->inv_n : a.inv_n);
      }
    }
    float sdf2[2];  // what the scatter multiplies the staged weights with: d loss / d f (BCE build), d y / d f (eikonal build)
    if (!EIK) {
    // ================================================================ phase 4: backward through the decoder
    float d2[8], d1[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 dw = splat2(delta) * (f32x2){w3r[2 * j], w3r[2 * j + 1]};
      d2[2 * j] = h2[2 * j] > 0.f ? dw[0] : 0.f;
      d2[2 * j + 1] = h2[2 * j + 1] > 0.f ? dw[1] : 0.f;
      dw3c[j] = pk_fma(splat2(delta), (f32x2){h2[2 * j], h2[2 * j + 1]}, dw3c[j]);
    }
    if (g == 0) db3 += delta;
    f32x4 e1[2] = {zero4(), zero4()}, e0 = zero4();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int m = 0; m < 2; ++m) e1[m] = mfma16(opa[V3_OPA2T + (8 * m + ks) * 64], d2[ks], e1[m]);
#pragma unroll
    for (int r = 0; r < 8; ++r) d1[r] = h1[r] > 0.f ? e1[r >> 2][r & 3] : 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) e0 = mfma16(opa[V3_OPA1T + ks * 64], d1[ks], e0);
    sdf2[0] = e0[0], sdf2[1] = e0[1];  // d loss / d f for features 2g, 2g+1 of this lane's point
    __builtin_amdgcn_sched_barrier(0);  // phase boundary: no operand of the next phase is fetched early
    SHINE_STAMP(3)  // loss + decoder backward

    // ================================================================ phase 5: decoder weight grads (transposed MFMA)
    if (a.decoder_grad_on) {
      const int i16 = lane & 15;  // operand role: row / column i16 = lane & 15, points 4 kk .. 4 kk + 3 (kk = lane >> 4)
#pragma unroll
      for (int r = 0; r < 8; ++r) {  // channel 16 (r >> 2) + 4 g + (r & 3)
        t_wr[(16 * (r >> 2) + (r & 3)) * V3_TT] = d2[r];
        t_wr[(32 + 16 * (r >> 2) + (r & 3)) * V3_TT] = h1[r];
      }
      wave_lds_fence();
      {
        float4 la[2], lb[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          la[m] = *reinterpret_cast<const float4*>(t_rd + (16 * m) * V3_TT);
          lb[m] = *reinterpret_cast<const float4*>(t_rd + (32 + 16 * m) * V3_TT);
          db2acc[m] += (la[m].x + la[m].y) + (la[m].z + la[m].w);  // db2 rides on the transposed operands
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) {  // dW2[out][in] += d2[out][k] h1[in][k]
            accW2[m][n] = mfma16(la[m].x, lb[n].x, accW2[m][n]);
            accW2[m][n] = mfma16(la[m].y, lb[n].y, accW2[m][n]);
            accW2[m][n] = mfma16(la[m].z, lb[n].z, accW2[m][n]);
            accW2[m][n] = mfma16(la[m].w, lb[n].w, accW2[m][n]);
          }
      }
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 8; ++r) t_wr[(16 * (r >> 2) + (r & 3)) * V3_TT] = d1[r];
      f_wr[32 * V3_TT] = f2[0];
      f_wr[33 * V3_TT] = f2[1];
      wave_lds_fence();
      {
        float4 la[2];
        // B columns 0..7 = f, column 8 = ones: accW1[:, 8] accumulates db1 = sum_k d1[ch][k] in the spare MFMA lanes
        float4 lb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i16 < F) lb = *reinterpret_cast<const float4*>(t_rd + 32 * V3_TT);
        if (i16 == F) lb = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
        for (int m = 0; m < 2; ++m) la[m] = *reinterpret_cast<const float4*>(t_rd + (16 * m) * V3_TT);
#pragma unroll
        for (int m = 0; m < 2; ++m) {  // dW1[ch][feat] += d1[ch][k] f[feat][k]
          accW1[m] = mfma16(la[m].x, lb.x, accW1[m]);
          accW1[m] = mfma16(la[m].y, lb.y, accW1[m]);
          accW1[m] = mfma16(la[m].z, lb.z, accW1[m]);
          accW1[m] = mfma16(la[m].w, lb.w, accW1[m]);
        }
      }
      wave_lds_fence();
    }
    __builtin_amdgcn_sched_barrier(0);  // phase boundary: no operand of the next phase is fetched early
    SHINE_STAMP(5)  // weight grads

    } else {
    // ================================================================ phase 4 (eikonal build): closed form, SURVEY.md §8a
    // The decoder has ONE output, so everything the loss sends back is the eikonal chain scaled by the point's delta:
    //   d2 = delta v2, d1 = delta v1, d loss_bce / d f = delta J     (v2 = m2 .* w3, v1 = m1 .* W2^T v2, J = W1^T v1 = dy/df)
    // and the weight grads of both terms contract in ONE pass per matrix:
    //   dW2 += v2 (x) (delta h1 + a1),  dW1 += v1 (x) (delta f + r),  db2 += sum delta v2,  db1 += sum delta v1.
    if (g == 0) db3 += delta;
    float v2[8], v1[8], a1[8], J2[2], r2[2], qv[3] = {0.f, 0.f, 0.f};
    {
      f32x4 ev[2] = {zero4(), zero4()}, ej = zero4();
#pragma unroll
      for (int r = 0; r < 8; ++r) v2[r] = h2[r] > 0.f ? w3r[r] : 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int m = 0; m < 2; ++m) ev[m] = mfma16(opa[V3_OPA2T + (8 * m + ks) * 64], v2[ks], ev[m]);
#pragma unroll
      for (int r = 0; r < 8; ++r) v1[r] = h1[r] > 0.f ? ev[r >> 2][r & 3] : 0.f;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) ej = mfma16(opa[V3_OPA1T + ks * 64], v1[ks], ej);
      J2[0] = ej[0], J2[1] = ej[1];  // d y / d f_{2g}, d y / d f_{2g+1}
    }
    float gx[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {  // get_gradient(coord, pred) * sigma   (utils/tools.py:175-185, shine_batch.py:141-142)
      float sm = fmaf(J2[1], A2[1][e], J2[0] * A2[0][e]);
      sm = xsum32(sm, sm);  // all-reduce over the point's four lanes
      sm = xsum16(sm, sm);
      gx[e] = a.sigma * sm;
    }
    if (valid && g == 0 && a.grad_x) {
      a.grad_x[3 * po] = gx[0];
      a.grad_x[3 * po + 1] = gx[1];
      a.grad_x[3 * po + 2] = gx[2];
    }
    if (EXT) {  // d loss / d g from autograd
      if (valid) {
        qv[0] = a.ext_q[3 * po];
        qv[1] = a.ext_q[3 * po + 1];
        qv[2] = a.ext_q[3 * po + 2];
      }
    } else if (valid && wgt > 0.f) {  // surface samples only (shine_batch.py:137,183)
      const float gn = sqrtf(gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2]);
      const float ee = 1.0f - gn;
      if (g == 0) eik_acc += ee * ee;
      const float coef = gn > 0.f ? (-2.0f * ee / gn) * (a.weight_e * inv_nsurf) : 0.f;  // norm's sub-gradient 0 at 0
      qv[0] = coef * gx[0];
      qv[1] = coef * gx[1];
      qv[2] = coef * gx[2];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) r2[t] = a.sigma * (A2[t][0] * qv[0] + A2[t][1] * qv[1] + A2[t][2] * qv[2]);
    {
      f32x4 t1[2] = {zero4(), zero4()}, t2[2] = {zero4(), zero4()};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int m = 0; m < 2; ++m) t1[m] = mfma16(opa[V3_OPA1 + (2 * m + t) * 64], r2[t], t1[m]);
#pragma unroll
      for (int r = 0; r < 8; ++r) a1[r] = h1[r] > 0.f ? t1[r >> 2][r & 3] : 0.f;  // (W1 r) .* m1
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int m = 0; m < 2; ++m) t2[m] = mfma16(opa[V3_OPA2 + (8 * m + ks) * 64], a1[ks], t2[m]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x2 a2;  // (W2 a1) .* m2
        a2[0] = h2[2 * j] > 0.f ? t2[j >> 1][2 * (j & 1)] : 0.f;
        a2[1] = h2[2 * j + 1] > 0.f ? t2[j >> 1][2 * (j & 1) + 1] : 0.f;
        const f32x2 dl = splat2(delta);
        dw3c[j] += pk_fma(dl, (f32x2){h2[2 * j], h2[2 * j + 1]}, a2);
        db2c[j] = pk_fma(dl, (f32x2){v2[2 * j], v2[2 * j + 1]}, db2c[j]);
        db1c[j] = pk_fma(dl, (f32x2){v1[2 * j], v1[2 * j + 1]}, db1c[j]);
      }
    }
    sdf2[0] = J2[0], sdf2[1] = J2[1];
    __builtin_amdgcn_sched_barrier(0);  // phase boundary
    SHINE_STAMP(3)  // loss + decoder backward (eikonal chain)

    // ================================================================ phase 5 (eikonal build): decoder weight grads
    if (a.decoder_grad_on) {
      const int i16 = lane & 15;
      f32x2 dha[4];  // delta h1 + a1
#pragma unroll
      for (int j = 0; j < 4; ++j) dha[j] = pk_fma(splat2(delta), (f32x2){h1[2 * j], h1[2 * j + 1]}, (f32x2){a1[2 * j], a1[2 * j + 1]});
#pragma unroll
      for (int r = 0; r < 8; ++r) {  // channel 16 (r >> 2) + 4 g + (r & 3)
        t_wr[(16 * (r >> 2) + (r & 3)) * V3_TT] = v2[r];
        t_wr[(32 + 16 * (r >> 2) + (r & 3)) * V3_TT] = dha[r >> 1][r & 1];
      }
      wave_lds_fence();
      {
        float4 la[2], lb[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          la[m] = *reinterpret_cast<const float4*>(t_rd + (16 * m) * V3_TT);
          lb[m] = *reinterpret_cast<const float4*>(t_rd + (32 + 16 * m) * V3_TT);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) {  // dW2[out][in] += v2[out][k] (delta h1 + a1)[in][k]
            accW2[m][n] = mfma16(la[m].x, lb[n].x, accW2[m][n]);
            accW2[m][n] = mfma16(la[m].y, lb[n].y, accW2[m][n]);
            accW2[m][n] = mfma16(la[m].z, lb[n].z, accW2[m][n]);
            accW2[m][n] = mfma16(la[m].w, lb[n].w, accW2[m][n]);
          }
      }
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 8; ++r) t_wr[(16 * (r >> 2) + (r & 3)) * V3_TT] = v1[r];
      f_wr[32 * V3_TT] = fmaf(delta, f2[0], r2[0]);
      f_wr[33 * V3_TT] = fmaf(delta, f2[1], r2[1]);
      wave_lds_fence();
      {
        float4 la[2];
        float4 lb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i16 < F) lb = *reinterpret_cast<const float4*>(t_rd + 32 * V3_TT);
#pragma unroll
        for (int m = 0; m < 2; ++m) la[m] = *reinterpret_cast<const float4*>(t_rd + (16 * m) * V3_TT);
#pragma unroll
        for (int m = 0; m < 2; ++m) {  // dW1[ch][feat] += v1[ch][k] (delta f + r)[feat][k]
          accW1[m] = mfma16(la[m].x, lb.x, accW1[m]);
          accW1[m] = mfma16(la[m].y, lb.y, accW1[m]);
          accW1[m] = mfma16(la[m].z, lb.z, accW1[m]);
          accW1[m] = mfma16(la[m].w, lb.w, accW1[m]);
        }
      }
      wave_lds_fence();
    }
    // weights for the scatter, staged now that delta and q are known: row (level, corner) of this point receives
    //   (delta w_c + sigma (d w_c / d x . q)) J      (BCE part delta J w_c, eikonal part sigma (dw_c/dx . q) J);
    // a miss stages 0 (its eikonal terms would all land on the trash row, where they cancel: sum_c dw_c/dx = 0)
#pragma unroll
    for (int cp = 0; cp < 4; ++cp) {  // corners (2 cp, 2 cp + 1), packed
      f32x2 dw[3];
      corner_dw_pair(X, Y, Z, cp, dw);
      f32x2 cq = dw[0] * splat2(qv[0]);
      cq = pk_fma(dw[1], splat2(qv[1]), cq);
      cq = pk_fma(dw[2], splat2(qv[2]), cq);
      const f32x2 wp = {w[2 * cp], w[2 * cp + 1]};
      const f32x2 o = pk_fma(splat2(delta), wp, splat2(a.sigma) * cq);  // a miss stages 0: w = 0 and dt = 0
      st_w[(2 * cp) * V3_WP] = o[0];
      st_w[(2 * cp + 1) * V3_WP] = o[1];
    }
    if (g == 0) R2[V3_DL + o_pt] = delta;  // the trash rows need delta J (their weights sum to 1)
    __builtin_amdgcn_sched_barrier(0);  // phase boundary
    SHINE_STAMP(5)  // weight grads
    }
    // ================================================================ phase 6: feature-grad scatter (run-length)
    f_wr[V3_DF] = sdf2[0];
    f_wr[V3_DF + V3_DFP] = sdf2[1];
    wave_lds_fence();
    {
      // this lane's trash level: the points that miss level sc
      const unsigned int mymiss = sc < L ? (~(unsigned int)(hit64 >> (16 * (sc & 3))) & validmask) : 0u;
      float dfr[V3_TP];
#pragma unroll
      for (int j = 0; j < V3_TP / 4; ++j) {
        const float4 v = *reinterpret_cast<const float4*>(sc_df + 4 * j);
        dfr[4 * j] = v.x;
        dfr[4 * j + 1] = v.y;
        dfr[4 * j + 2] = v.z;
        dfr[4 * j + 3] = v.w;
      }
      // staged operands of one level: weights and ids of this lane's corner for the 16 points (8 x 16-B LDS reads).
      // Level s + 1's are requested BEFORE level s is walked: the walk is a chain of scalar branches (basic blocks the
      // scheduler cannot move loads across), and an LDS round trip per level was exposed (~150 cycles, 8 x per tile).
      float4 wq[2][V3_TP / 4];
      int4 iq[2][V3_TP / 4];
#pragma unroll
      for (int j = 0; j < V3_TP / 4; ++j) {
        wq[0][j] = *reinterpret_cast<const float4*>(sc_w + 4 * j);
        iq[0][j] = *reinterpret_cast<const int4*>(sc_ids + 4 * j);
      }
      {  // trash rows: the plain sum of d loss_bce / d f over the misses (the 8 corner weights of a missed node sum to 1)
        float dl[EIK ? V3_TP : 1];
        if (EIK) {  // the staged vector is J: d loss_bce / d f = delta J
#pragma unroll
          for (int j = 0; j < V3_TP / 4; ++j) {
            const float4 v = *reinterpret_cast<const float4*>(R2 + V3_DL + 4 * j);
            dl[4 * j] = v.x, dl[4 * j + 1] = v.y, dl[4 * j + 2] = v.z, dl[4 * j + 3] = v.w;
          }
        }
#pragma unroll
        for (int p2 = 0; p2 < V3_TP; ++p2) {
          const unsigned int keep = 0u - ((mymiss >> p2) & 1u);
          const float t = EIK ? dfr[p2] * dl[p2] : dfr[p2];
          trash_sum += __uint_as_float(__float_as_uint(t) & keep);
        }
      }
#pragma unroll
      for (int s = 0; s < L; ++s) {
        if (s + 1 < L) {
#pragma unroll
          for (int j = 0; j < V3_TP / 4; ++j) {
            wq[(s + 1) & 1][j] = *reinterpret_cast<const float4*>(sc_w + ((s + 1) * 8) * V3_WP + 4 * j);
            iq[(s + 1) & 1][j] = *reinterpret_cast<const int4*>(sc_ids + ((s + 1) * 8) * V3_WP + 4 * j);
          }
        }
        float* gbase = SLICED ? sl->grad[s] : a.lv[s].grad;
        if (gbase) {
          float wr[V3_TP];
          int idr[V3_TP];
#pragma unroll
          for (int j = 0; j < V3_TP / 4; ++j) {
            const float4 v = wq[s & 1][j];
            const int4 u = iq[s & 1][j];
            wr[4 * j] = v.x, wr[4 * j + 1] = v.y, wr[4 * j + 2] = v.z, wr[4 * j + 3] = v.w;
            idr[4 * j] = u.x, idr[4 * j + 1] = u.y, idr[4 * j + 2] = u.z, idr[4 * j + 3] = u.w;
          }
          int rid = run_id[s], rhit = run_hit[s];
          float racc = run_acc[s];
          int lt = -1;     // FAR: the lattice entry of the open run, read when the run starts / is carried into this tile
          float lv = 0.f;
          float* const lat = cbase + s * V3_CLEVEL;
          if (FAR && rhit) lattice_peek(lat, rid, sq, lt, lv);
          const unsigned int cm = (unsigned int)(chg64 >> (16 * s)) & 0xFFFFu;
          const unsigned int hm = (unsigned int)(hit64 >> (16 * s)) & 0xFFFFu;
#pragma unroll
          for (int p2 = 0; p2 < V3_TP; ++p2) {
            if (cm & (1u << p2)) {  // a new node (or a run of misses) starts here: close the open run
              if (rhit) {  // scalar branch
                if (FAR) lattice_close(lat, gbase, rid, sq, racc, lt, lv);
                else grad_atomic(gbase + (unsigned int)rid, racc);
              }
              racc = 0.f;
              // float offset of this lane's (corner row, feature); FAR: the staged id itself (row | lattice slot << 25)
              rid = FAR ? idr[p2] : (idr[p2] << 3) | sq;
              rhit = (int)((hm >> p2) & 1u);
              if (FAR && rhit) lattice_peek(lat, rid, sq, lt, lv);
              // the touched-row flags (unique(hierarchical_indices) without -1, for shine_regularize) are set here, at the run
              // start of every hit node, by one lane per corner
              if (MARK && rhit && SL_TOUCHED(s) && sq == 0)
                SL_TOUCHED(s)[FAR ? idr[p2] & ((1 << V3_ROW_BITS) - 1) : idr[p2]] = 1;
            }
            racc = fmaf(wr[p2], dfr[p2], racc);  // misses and padding lanes staged w = 0
          }
          run_id[s] = rid;
          run_hit[s] = rhit;
          run_acc[s] = racc;
        }
      }
    }
    wave_lds_fence();
    if (SHINE_V3_ABLATE & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (measurement: wait for the atomics' acknowledgements)
    SHINE_STAMP(4)  // scatter
  }

  // ---- end of the wave's run: flush the open node runs
#pragma unroll
  for (int s = 0; s < L; ++s) {
    float* gbase = SLICED ? sl->grad[s] : a.lv[s].grad;
    if (!gbase) continue;
    if (FAR) {
      float* const lat = cbase + s * V3_CLEVEL;
      if (run_hit[s]) {
        int lt;
        float lv;
        lattice_peek(lat, run_id[s], sq, lt, lv);
        lattice_close(lat, gbase, run_id[s], sq, run_acc[s], lt, lv);
      }
      // what the lattice still holds goes to memory: lane group sc flushes slots 8 sc .. 8 sc + 7
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int slot = 8 * sc + j;
        const int T = reinterpret_cast<const int*>(lat)[slot];
        if (T >= 0) grad_atomic(gbase + ((unsigned int)T << 3) + sq, lat[V3_CSLOTS + slot * F + sq]);
      }
    } else if (run_hit[s]) {
      grad_atomic(gbase + (unsigned int)run_id[s], run_acc[s]);
    }
  }
  __syncthreads();  // every wave is done with its staging region: it now holds the wave's partial vector
  float* wvec = sm.wave[wv];
  if (sc < L) wvec[PART_TRASH + sc * 8 + sq] = trash_sum;
  if (a.decoder_grad_on) {
    const int jc = lane & 15, rr = lane >> 4;  // accumulator role: column jc, rows 4 rr + r
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * m + 4 * rr + r;
        wvec[MLP_W2 + row * H + jc] = accW2[m][0][r];
        wvec[MLP_W2 + row * H + 16 + jc] = accW2[m][1][r];
        if (jc < F || (!EIK && jc == F)) wvec[jc < F ? MLP_W1 + row * F + jc : MLP_B1 + row] = accW1[m][r];  // BCE: column 8 of accW1 is db1
        const float w3v = row16_sum(dw3c[2 * m + (r >> 1)][r & 1]);  // channel 16 m + 4 g + r over the 16 points of the DPP row
        if (pt == 0) wvec[MLP_W3 + 16 * m + 4 * g + r] = w3v;
        if (EIK) {
          const float b2v = row16_sum(db2c[2 * m + (r >> 1)][r & 1]), b1v = row16_sum(db1c[2 * m + (r >> 1)][r & 1]);
          if (pt == 0) {
            wvec[MLP_B2 + 16 * m + 4 * g + r] = b2v;
            wvec[MLP_B1 + 16 * m + 4 * g + r] = b1v;
          }
        }
      }
    if (!EIK) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {  // channel 16 m + (lane & 15), the four point groups kk
        float v = db2acc[m];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 16) wvec[MLP_B2 + 16 * m + lane] = v;
      }
    }
    const float b3v = wave_sum(db3);
    if (lane == 0) wvec[MLP_B3] = b3v;
  }
  {
    const double ls = wave_sum_d((double)loss_acc), cs = wave_sum_d((double)cnt_acc);
    const double es = EIK ? wave_sum_d((double)eik_acc) : 0.0;
    if (lane == 0) {
      atomicAdd(&s_loss[0], ls);
      atomicAdd(&s_loss[1], cs);
      if (EIK) atomicAdd(&s_loss[2], es);
    }
  }
  SHINE_STAMP(6)  // flush
  __syncthreads();
  SHINE_STAMP(7)  // wait for the workgroup
  if (PROF && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a.prof[wave_g * 8 + k] = pc[k];
  }
  const int mlp_lo = a.decoder_grad_on ? 0 : SHINE_MLP_PARAMS;  // a frozen decoder has no sums to move
  float* dst = a.partials + (long long)(SLICED ? sl->part_bid : bid) * PART_STRIDE;
  for (int idx = tid; idx < PART_TRASH + L * 8; idx += NT) {
    float v = 0.f;
    if (idx >= mlp_lo) {
#pragma unroll
      for (int w = 0; w < WAVES; ++w) v += sm.wave[w][idx];
    }
    dst[idx] = v;
  }
  for (int idx = PART_TRASH + L * 8 + tid; idx < PART_FLOATS; idx += NT) dst[idx] = 0.f;
  if (tid == 0) {
    double* dl = reinterpret_cast<double*>(dst + PART_LOSS);
    dl[0] = s_loss[0];
    dl[1] = s_loss[1];
    dl[2] = s_loss[2];
  }
  // cfg->defer_reduce: no reduction launch follows, so the iteration hooks (optimiser step count, regulariser accumulator)
  // ride here — nothing in this launch reads either
  if (a.defer_reduce && bid == 0) {
    if (tid == 0 && a.adam_state) adam_advance(a.adam_state, a.adam_b1, a.adam_b2);
    if (tid == 64 && a.zero_f64) *a.zero_f64 = 0.0;
  }
#undef SHINE_STAMP
#undef SL_N
#undef SL_PERM
#undef SL_TOUCHED
}

}  // namespace shine
