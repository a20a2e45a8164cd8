// shine_step_v5.hip — the fused SHINE training step with ROLE-SPECIALISED waves (16-point tiles, planned / pool batches).
//
//   query    FeatureOctree.query_feature   model/feature_octree.py:199-244
//   decode   Decoder.sdf                   model/decoder.py:49-63
//   loss     sdf_bce_loss                  utils/loss.py:17-24
//   backward cur_loss.backward()           shine_batch.py:208-209 (closed form, SURVEY.md §8a math contract)
//
// Same arithmetic, staging layouts and outputs as shine_step_v3.hip — every phase below is that kernel's phase, on the same
// lane roles.  What changes is WHO runs a phase.  In k_step_v3 one wave walks a tile through query -> decoder -> scatter, so
// its three dependent memory round trips (slot -> corner ids -> rows), its MFMA chain and its atomics are one serial chain;
// with two 186-VGPR waves per SIMD the counters show the waves parked at s_waitcnt 48 % of their cycles and the fp32
// datapath (MFMA + VALU, one datapath: tools/ubench/mfma_valu_overlap.hip) 39 % occupied (profiles/r02_pmc_*).
// Here a 1024-thread workgroup is 4 pipelines (one per SIMD: waves w, w+4, w+8, w+12 share a SIMD) of 4 waves each:
//
//   G  query/gather   lane = (point, LEVEL).  Software-pipelined loads — rows of tile j, corner ids of j+1, point data of
//                     j+2, sample index of j+3 all issued together at the top of the iteration, so ONE round trip per tile
//                     is exposed instead of three — smooth-step weights, weighted row sum, reduce-scatter
//                     (v_permlane32/16_swap); stages ids / weights / features / masks in an LDS ring slot.
//   D0, D1 decoder    lane = (point, channel group).  Even / odd tiles: exact-fp32 MFMA chain forward, loss, backward,
//                     weight grads in register accumulators (as k_step_v3 phases 2-5); writes d loss / d f into the slot.
//   S  scatter        lane = (corner, feature).  Run-length walk + one 64-lane atomic per node run (phase 6), in order
//                     over the pipeline's contiguous tile range (node runs carry across tiles).
//
// Every wave needs <= 128 VGPRs (its own role's state only), so 4 waves per SIMD are resident and a wave that waits for
// memory or for an MFMA result leaves its SIMD to three others doing a different kind of work.  Hand-off: single-writer
// sequence counters in LDS (g_pub, d_pub[2], s_done per pipeline), polled with ds_read + s_sleep; DS operations of one
// wave execute in order, so {data writes, s_waitcnt lgkmcnt(0), counter write} publishes a slot without any barrier.
// Every spin is bounded (a broken hand-off poisons the loss with NaN instead of hanging the GPU).
#include "../shine_tile16.hpp"

namespace shine {

constexpr int V5_PIPES = 4;   // pipelines per workgroup = SIMDs per CU
// waves of a pipeline: ND decoder waves (tiles j = d mod ND), one gather wave, one scatter wave.  ND = 2 (16 waves per CU,
// <= 128 VGPRs) for the BCE build; the eikonal build's decoder wave needs more registers: ND = 1 (12 waves, <= 168 VGPRs).
// ring slot (floats).  ids / w / df as in k_step_v3's per-wave staging region; f = the reduce-scattered features in
// (point, g) lane order; meta = label[16] weight[16] pidx[16] | chg64 hit64 validmask.  Eikonal build: + the three axis
// weights / derivatives of every (point, level) lane and A = d f / d x of every (point, g) lane.
constexpr int S5_IDS = 0;
constexpr int S5_W = S5_IDS + LCAP * 8 * V3_WP;
constexpr int S5_F = S5_W + LCAP * 8 * V3_WP;
constexpr int S5_DF = S5_F + 128;
constexpr int S5_META = S5_DF + 8 * V3_DFP + 16;
constexpr int S5_BCE_FLOATS = S5_META + 64;
constexpr int S5_AX = S5_BCE_FLOATS;
constexpr int S5_A2 = S5_AX + 6 * 64;
constexpr int S5_EIK_FLOATS = S5_A2 + 6 * 64;
#ifndef SHINE_V5_SLEEP  // s_sleep argument between two polls of a hand-off counter (units of 64 clocks)
#define SHINE_V5_SLEEP 2
#endif
#ifndef SHINE_V5_SPIN
#define SHINE_V5_SPIN (1 << 21)
#endif
#ifndef SHINE_V5_PRIO  // wave priorities by role: scatter 3 > gather 2 > decoder 0 (the scatter wave is the youngest of its
#define SHINE_V5_PRIO 1  // SIMD and its dependent v_fmac chain queues behind the decoder waves' MFMAs)
#endif
#ifndef SHINE_V5_CH  // tiles per chunk of the interleaved tile assignment (0: one contiguous range per pipeline)
#define SHINE_V5_CH 4
#endif
#ifndef SHINE_V5_DYN  // 1: the chunks of a workgroup (b, b + B, b + 2 B, ...) are CLAIMED by its four gather waves from an LDS
#define SHINE_V5_DYN 1  // counter instead of being dealt round-robin: a pipeline that drew expensive tiles simply claims fewer
#endif
constexpr int V5_CH = SHINE_V5_CH;
#ifndef SHINE_V5_PREFIX  // 1: prefix-sum scatter (scatter_level_prefix, shine_tile16.hpp) instead of the serial walk
#define SHINE_V5_PREFIX 0
#endif
#ifndef SHINE_V5_ABL  // measurement builds only: 1 no atomics, 2 no weight-grad phase, 4 no scatter walk, 8 no row gathers
#define SHINE_V5_ABL 0
#endif
#ifndef SHINE_V5_PROF  // measurement builds only (tools/mk_variant.py): per-wave cycle counters through a.prof
#define SHINE_V5_PROF 0
#endif

enum { SY_GPUB = 0, SY_DPUB0 = 1, SY_DPUB1 = 2, SY_SDONE = 3, SY_ERR = 4, SY_TOTAL = 5, SY_WORDS = 8 };

// The hand-off counters are read and written through explicit LDS (address space 3) pointers: behind a generic pointer the
// compiler emits flat_load / flat_store, whose completion it tracks with vmcnt — every poll would drain the wave's
// outstanding gathers and atomics.
typedef volatile int __attribute__((address_space(3))) lds_vint;
__device__ __forceinline__ lds_vint* lds_word(int* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (lds_vint*)p;
#pragma clang diagnostic pop
}
typedef const void __attribute__((address_space(1))) gvoid;
typedef void __attribute__((address_space(3))) lvoid;
__device__ __forceinline__ gvoid* to_global(const float* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (gvoid*)p;
#pragma clang diagnostic pop
}
__device__ __forceinline__ lvoid* to_lds(float* p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  return (lvoid*)p;
#pragma clang diagnostic pop
}
// wait until the single-writer sequence counter reaches `target` (bounded: sets the error word and goes on)
__device__ __forceinline__ void wait_ge(int* flag, int target, int* err, long long& waited) {
  const long long t0 = SHINE_V5_PROF ? clk() : 0;
  lds_vint* const f = lds_word(flag);
  int spins = 0;
  for (;;) {
    const int v = __builtin_amdgcn_readfirstlane(*f);
    if (v >= target) break;
    __builtin_amdgcn_s_sleep(SHINE_V5_SLEEP);
    if (++spins > SHINE_V5_SPIN) {
      *lds_word(err) = 1;
      break;
    }
  }
  asm volatile("" ::: "memory");
  if (SHINE_V5_PROF) waited += clk() - t0;
}
// the same for a consumer of tiles: false when tile j does not exist (the gather wave publishes the pipeline's tile count in
// `total` once it knows it; INT_MAX until then)
__device__ __forceinline__ bool wait_tile(int* flag, int* total, int j, int* err, long long& waited) {
  const long long t0 = SHINE_V5_PROF ? clk() : 0;
  lds_vint* const f = lds_word(flag);
  lds_vint* const t = lds_word(total);
  int spins = 0;
  bool there = true;
  for (;;) {
    if (__builtin_amdgcn_readfirstlane(*f) >= j + 1) break;
    if (__builtin_amdgcn_readfirstlane(*t) <= j) {
      there = false;
      break;
    }
    __builtin_amdgcn_s_sleep(SHINE_V5_SLEEP);
    if (++spins > SHINE_V5_SPIN) {
      *lds_word(err) = 1;
      there = false;
      break;
    }
  }
  asm volatile("" ::: "memory");
  if (SHINE_V5_PROF) waited += clk() - t0;
  return there;
}
// publish: every DS write of this wave so far is performed before the counter moves
__device__ __forceinline__ void publish(int* flag, int value) {
  wave_lds_fence();
  *lds_word(flag) = value;
}

template <int L, bool EIK, int ND>
__global__ __launch_bounds__((ND + 2) * V5_PIPES * 64) void k_step_v5(V1Args a) {
  constexpr int V5_NT = (ND + 2) * V5_PIPES * 64;
  // partial sums at the flush: one decoder-gradient vector per decoder wave, then the trash-row sums of the scatter waves
  constexpr int DVEC = (SHINE_MLP_PARAMS + LCAP * 8 + 3) / 4 * 4, N_DVEC = V5_PIPES * ND;  // decoder grads + trash rows
  constexpr int K = EIK ? 2 : 3;  // ring depth (what the 160 KB of LDS leave room for)
  constexpr int SLOT = EIK ? S5_EIK_FLOATS : S5_BCE_FLOATS;
  static_assert(N_DVEC * DVEC <= V5_PIPES * K * SLOT, "the flush re-uses the ring for the partial sums");
  __shared__ float s_opA[V3_OPTOTAL];
  __shared__ float s_bias[100];
  __shared__ double s_loss[4];
  __shared__ int s_sync[V5_PIPES][SY_WORDS];
  __shared__ int s_claim;  // SHINE_V5_DYN: chunks of this workgroup handed out so far
  __shared__ float s_ring[V5_PIPES * K * SLOT];
  __shared__ float s_r2[V5_PIPES * ND][V3_R2];  // transpose scratch of the decoder waves
  __shared__ float s_dma[V5_PIPES][8 * 256];  // landing zone of the gather wave's LDS-DMA rows: [corner 4..7][half][lane] x 16 B

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pl = wv & 3, role = wv >> 2;  // role < ND: decoder (tiles role mod ND), ND: gather, ND + 1: scatter
  const int pt = lane & 15, g = lane >> 4;
  const bool poly = a.poly != 0;
  long long waited = 0;  // measurement builds: cycles this wave spent polling a hand-off counter
  const long long t_start = SHINE_V5_PROF ? clk() : 0;

  load_decoder_operands<V5_NT>(a, s_opA, s_bias, tid);
  if (tid < V5_PIPES * SY_WORDS) (&s_sync[0][0])[tid] = (tid % SY_WORDS) == SY_TOTAL ? 0x7fffffff : 0;
  if (tid == 0) {
    s_loss[0] = s_loss[1] = s_loss[2] = s_loss[3] = 0.0;
    s_claim = 0;
  }
  __syncthreads();

  // Tiles.  V5_CH == 0: workgroup b owns [b T / B, (b + 1) T / B), its pipelines contiguous quarters of that.
  // V5_CH > 0: the stream is cut into chunks of V5_CH tiles dealt round-robin to the 4 B pipelines of the launch — every
  // pipeline gets a sample of the whole map instead of one neighbourhood, which evens out what contiguous ranges do not:
  // ranges of free-space samples (all misses: nothing to gather or scatter) next to ranges on surfaces (a node run every
  // few points).  Node runs restart at chunk borders (the gather wave forgets its last node there).
  int begin = 0;  // (positions fit 32 bits: shine_train_step_v5 refuses launches of 2^31 points or more)
  const int end = (int)a.n;
  int njobs = 0;  // static assignments: tiles of this pipeline (dynamic: known to the gather wave at the end only)
  int tiles_done = 0;
  constexpr int CHD = V5_CH > 0 ? V5_CH : 1;
  static_assert(!SHINE_V5_DYN || V5_CH >= 4, "claiming one chunk ahead covers the gather wave's three-tile look-ahead");
  const long long n_chunks_all = (a.tiles + CHD - 1) / CHD;
  const long long n_pipes = (long long)gridDim.x * V5_PIPES, pid = (long long)blockIdx.x * V5_PIPES + pl;
  if (V5_CH == 0) {
    const long long t0 = ((long long)blockIdx.x * a.tiles) / gridDim.x, t1 = ((long long)(blockIdx.x + 1) * a.tiles) / gridDim.x;
    const long long nt = t1 - t0;
    const long long lo = t0 + (pl * nt) / V5_PIPES, hi = t0 + ((pl + 1) * nt) / V5_PIPES;
    begin = (int)(V3_TP * lo);
    njobs = (int)(hi - lo);
  } else {
    constexpr int CH = V5_CH > 0 ? V5_CH : 1;
    const long long n_chunks = (a.tiles + CH - 1) / CH;
    const long long mine = pid < n_chunks ? (n_chunks - pid + n_pipes - 1) / n_pipes : 0;  // chunks pid, pid + P, ...
    njobs = (int)(mine * V5_CH);
    if (mine > 0) {  // the stream's last chunk may be short
      const long long last = pid + (mine - 1) * n_pipes;
      const long long left = a.tiles - last * V5_CH;
      if (left < V5_CH) njobs -= (int)(V5_CH - left);
    }
  }
  // position in the visiting order of the first point of this pipeline's j-th tile
  auto tile_base = [&](int j) -> int {
    if (V5_CH == 0) return begin + V3_TP * j;
    const int c = (int)pid + (j / CHD) * (int)n_pipes;
    return V3_TP * (c * CHD + (j % CHD));
  };
  int* const sync = s_sync[pl];
  float* const ring = s_ring + pl * K * SLOT;
  const long long t_loop = SHINE_V5_PROF ? clk() : 0;

  // ------------------------------------------------------------------------------------------------ role state
  // decoder waves
  f32x4 accW2[2][2], accW1[2];
  float dw3c[8], db2acc[2] = {0.f, 0.f}, db2c[EIK ? 8 : 1], db1c[EIK ? 8 : 1];
  float db3 = 0.f, loss_acc = 0.f, eik_acc = 0.f;
  int cnt_acc = 0;
  float trash[LCAP][2];  // trash rows (misses): sum over this lane's points of d loss_bce / d f_{2g+t}, per level
#pragma unroll
  for (int s = 0; s < LCAP; ++s) trash[s][0] = trash[s][1] = 0.f;
  // scatter waves
  const int sq = lane & 7;

  if (role < ND) {
    // ============================================================================================ D: decoder waves
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      accW1[m] = zero4();
#pragma unroll
      for (int n = 0; n < 2; ++n) accW2[m][n] = zero4();
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) dw3c[r] = 0.f;
#pragma unroll
    for (int r = 0; r < (EIK ? 8 : 1); ++r) db2c[r] = db1c[r] = 0.f;
    float inv_nsurf = 0.f;
    if (EIK) {
      const long long ns = a.n_surf ? *a.n_surf : 0;
      inv_nsurf = ns > 0 ? 1.0f / (float)ns : 0.f;
    }
    const float b3 = s_bias[96];
    const float inv_sigma = 1.0f / a.sigma;
    const float4* sb4 = reinterpret_cast<const float4*>(s_bias);
    float* const R2 = s_r2[ND * pl + role];
    if (SHINE_V5_PRIO) __builtin_amdgcn_s_setprio(0);
    int lane_o = lane;
    int jk = role % K;  // slot of tile j = role, role + ND, ...
    for (int j = role;; j += ND) {
      if (!wait_tile(sync + SY_GPUB, sync + SY_TOTAL, j, sync + SY_ERR, waited)) break;
      ++tiles_done;
      asm volatile("" : "+v"(lane_o));  // opaque per tile: keeps pre-added LDS address variants out of loop-carried VGPRs
      const int o_pt = lane_o & 15, o_g = lane_o >> 4;
      float* const slot = ring + jk * SLOT;
      float* const t_wr = R2 + (4 * o_g) * V3_TT + o_pt;       // transpose writes: + (16 m + r) * V3_TT [+ 32 * V3_TT]
      const float* const t_rd = R2 + o_pt * V3_TT + 4 * o_g;   // operand reads (i16 = lane & 15, kk = lane >> 4)
      float* const f_wr = R2 + (2 * o_g) * V3_TT + o_pt;       // [feature 2g (+1)][pt] rows
      float* const df_wr = slot + S5_DF + (2 * o_g) * V3_DFP + o_pt;
      const float* const opa = s_opA + lane_o;
      // ---- inputs of this tile: features (2g, 2g+1) of point pt, the point's label / weight / output position
      const float2 fin = *reinterpret_cast<const float2*>(slot + S5_F + 2 * lane_o);
      const float f2[2] = {fin.x, fin.y};
      const float label = slot[S5_META + o_pt];
      const float wgt = slot[S5_META + 16 + o_pt];
      const int pidx = reinterpret_cast<const int*>(slot)[S5_META + 32 + o_pt];
      const bool valid = pidx >= 0;
      float A2[2][3];
      Axis X = {0.f, 0.f}, Y = {0.f, 0.f}, Z = {0.f, 0.f};
      bool lvhit = false;  // eikonal build: this lane as (point, LEVEL g) hit its node
      if (EIK) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int e = 0; e < 3; ++e) A2[t][e] = slot[S5_A2 + (3 * t + e) * 64 + lane_o];
        X.t = slot[S5_AX + 0 * 64 + lane_o], Y.t = slot[S5_AX + 1 * 64 + lane_o], Z.t = slot[S5_AX + 2 * 64 + lane_o];
        X.dt = slot[S5_AX + 3 * 64 + lane_o], Y.dt = slot[S5_AX + 4 * 64 + lane_o], Z.dt = slot[S5_AX + 5 * 64 + lane_o];
        lvhit = reinterpret_cast<const int*>(slot)[S5_META + 56 + o_g] >> o_pt & 1;  // hit bits of level g: 16-bit words
      }
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
      if (valid && g == 0 && a.pred) __builtin_nontemporal_store(y, a.pred + pidx);
      // ================================================================ phase 3: loss (hardware transcendentals, as v3)
      float delta = 0.f;
      {
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
          delta = lw * (sg - zt) * a.inv_n;
        }
      }
      float sdf2[2];
      if (!EIK) {
        // ============================================================== phase 4: backward through the decoder
        float d2[8], d1[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          d2[r] = h2[r] > 0.f ? delta * w3r[r] : 0.f;
          dw3c[r] = fmaf(delta, h2[r], dw3c[r]);
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
        // the scatter wave can start as soon as d loss / d f is staged: publish BEFORE the weight-grad phase
        df_wr[0] = sdf2[0];
        df_wr[V3_DFP] = sdf2[1];
        const int4 hitw = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(slot) + S5_META + 56);
        publish(sync + SY_DPUB0 + role, j + 1);
        {  // trash rows: a level the point misses sends d loss / d f there whole (the 8 corner weights of a node sum to 1)
          const int hw[4] = {hitw.x, hitw.y, hitw.z, hitw.w};
#pragma unroll
          for (int s = 0; s < L; ++s) {
            const bool miss = valid && !((hw[s] >> o_pt) & 1);
            trash[s][0] += miss ? sdf2[0] : 0.f;
            trash[s][1] += miss ? sdf2[1] : 0.f;
          }
        }
        // ============================================================== phase 5: decoder weight grads (transposed MFMA)
        if (a.decoder_grad_on && !(SHINE_V5_ABL & 2)) {
          const int i16 = lane & 15;
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
      } else {
        // ============================================================== phase 4 (eikonal build): closed form, SURVEY.md §8a
        //   d2 = delta v2, d1 = delta v1, d loss_bce / d f = delta J   (v2 = m2 .* w3, v1 = m1 .* W2^T v2, J = W1^T v1 = dy/df)
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
          a.grad_x[3 * (long long)pidx] = gx[0];
          a.grad_x[3 * (long long)pidx + 1] = gx[1];
          a.grad_x[3 * (long long)pidx + 2] = gx[2];
        }
        if (valid && wgt > 0.f) {  // surface samples only (shine_batch.py:137,183)
          const float gn = sqrtf(gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2]);
          const float ee = 1.0f - gn;
          if (g == 0) eik_acc += ee * ee;
          const float coef = gn > 0.f ? (-2.0f * ee / gn) * (a.weight_e * inv_nsurf) : 0.f;  // norm's sub-gradient 0 at 0
          qv[0] = coef * gx[0];
          qv[1] = coef * gx[1];
          qv[2] = coef * gx[2];
        }
        // the scatter's operands: staged vector J, staged weight (delta w_c + sigma (d w_c / d x . q)) of this lane AS
        // (point, level g), delta for the trash rows — all known now: publish before the second chain and the weight grads
        {
          float w[8];
          corner_weights(X.t, Y.t, Z.t, w);
          float* const st_w = slot + S5_W + (8 * o_g) * V3_WP + o_pt;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            float dwc[3];
            corner_dw(X, Y, Z, c, dwc);  // (a miss carries dt = 0 on every axis)
            const float cq = a.sigma * (dwc[0] * qv[0] + dwc[1] * qv[1] + dwc[2] * qv[2]);
            st_w[c * V3_WP] = lvhit ? fmaf(delta, w[c], cq) : 0.f;
          }
          df_wr[0] = J2[0];
          df_wr[V3_DFP] = J2[1];
          const int4 hitw = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(slot) + S5_META + 56);
          publish(sync + SY_DPUB0 + role, j + 1);
          const int hw[4] = {hitw.x, hitw.y, hitw.z, hitw.w};
#pragma unroll
          for (int s = 0; s < L; ++s) {  // trash rows: d loss_bce / d f = delta J (the eikonal terms cancel there)
            const bool miss = valid && !((hw[s] >> o_pt) & 1);
            trash[s][0] += miss ? delta * J2[0] : 0.f;
            trash[s][1] += miss ? delta * J2[1] : 0.f;
          }
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
          for (int r = 0; r < 8; ++r) {
            const float a2 = h2[r] > 0.f ? t2[r >> 2][r & 3] : 0.f;  // (W2 a1) .* m2
            dw3c[r] += fmaf(delta, h2[r], a2);
            db2c[r] = fmaf(delta, v2[r], db2c[r]);
            db1c[r] = fmaf(delta, v1[r], db1c[r]);
          }
        }
        // ============================================================== phase 5 (eikonal build): decoder weight grads
        if (a.decoder_grad_on) {
          const int i16 = lane & 15;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            t_wr[(16 * (r >> 2) + (r & 3)) * V3_TT] = v2[r];
            t_wr[(32 + 16 * (r >> 2) + (r & 3)) * V3_TT] = fmaf(delta, h1[r], a1[r]);
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
      }
      jk += ND;
      if (jk >= K) jk -= K;
    }
  } else if (role == ND) {
    // ============================================================================================ G: query / gather wave
    // lane (pt, g) owns LEVEL g of point pt (lanes of a level the tree does not have borrow the leaf level's pointers and
    // never hit)
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
    if (SHINE_V5_PRIO) __builtin_amdgcn_s_setprio(2);
    int last_slot = -2;  // this lane's level: the node of the previous tile's last point (carries runs across tiles)
    // Software pipeline over the dependent chain  position -> sample index -> {hash slot, coordinates} -> corner ids -> rows:
    // the rows of tile j + 1 are requested right after the rows of tile j have been consumed (the one point of an iteration
    // where everything in flight has landed), the corner ids of j + 1, the point data of j + 2 and the sample index of j + 3
    // at the top of iteration j — so the rest of an iteration and the first half of the next cover the rows' round trip.
    // Rows: corners 0-3 through registers, corners 4-7 global -> LDS directly (global_load_lds: no VGPRs).
    struct PD {
      int p;  // sample index (pool / batch), -1: no point
      float x0, x1, x2, label, weight;
      int slot;  // hash slot of this lane's level (-1: miss)
    };
    // dynamic assignment: the chunk this wave works on and the one it has claimed next (claimed one chunk ahead, because the
    // look-ahead of the load pipeline runs three tiles ahead of the tile being staged)
    int gc_cur = -1, gc_next = -1;
    int kcur = 0;
    const int n_chunks_i = (int)n_chunks_all, n_tiles_i = (int)a.tiles;
    auto claim = [&]() -> int {
      int v = 0;
      if (lane == 0) v = __hip_atomic_fetch_add(&s_claim, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      v = __builtin_amdgcn_readfirstlane(v);
      const long long gc = (long long)blockIdx.x + (long long)v * gridDim.x;
      return gc < n_chunks_i ? (int)gc : -1;
    };
    if (SHINE_V5_DYN) {
      gc_cur = claim();
      gc_next = claim();
    } else {
      *lds_word(sync + SY_TOTAL) = njobs;  // static assignment: the consumers know where the pipeline's stream ends
    }
    // first point (position in the visiting order) of this pipeline's j-th tile, or -1 when there is no such tile
    auto tile_pos = [&](int j) -> int {
      if (!SHINE_V5_DYN) return j < njobs ? tile_base(j) : -1;
      const int k = j / CHD, off = j % CHD;
      const int gc = k == kcur ? gc_cur : (k == kcur + 1 ? gc_next : -1);
      if (gc < 0) return -1;
      const int t = gc * CHD + off;
      return t < n_tiles_i ? V3_TP * t : -1;
    };
    auto pos = [&](int j) -> int {  // this lane's point of that tile; past the last tile: nothing to load
      const int b = tile_pos(j);
      return b >= 0 ? b + pt : end;
    };
    auto load_pm = [&](int i) -> int { return (a.perm && i < end) ? __builtin_nontemporal_load(a.perm + i) : 0; };
    auto load_pd = [&](int i, int pm) -> PD {  // i = position in the visiting order; pm = perm[i] if there is a perm
      PD d = {-1, 0.f, 0.f, 0.f, 0.f, 0.f, -1};
      if (i < end) {
        const long long p = a.perm ? pm : i;
        const long long si = a.pool_mode ? p : (long long)i;
        d.p = (int)p;
        if (lvl_on) d.slot = __builtin_nontemporal_load(a.slots + si * L + g);
        d.x0 = __builtin_nontemporal_load(a.coord + 3 * p);
        d.x1 = __builtin_nontemporal_load(a.coord + 3 * p + 1);
        d.x2 = __builtin_nontemporal_load(a.coord + 3 * p + 2);
        d.label = __builtin_nontemporal_load(a.label + p);
        if (EIK || a.weighted) d.weight = __builtin_nontemporal_load(a.weight + p);
      }
      return d;
    };
    auto load_ids = [&](int slot, int4& va, int4& vb) {
      const unsigned int sl = slot >= 0 ? (unsigned int)slot : 0u;
      va = lv_vals[2u * sl];
      vb = lv_vals[2u * sl + 1u];
    };
    float4 ra0[4], ra1[4];  // corner rows 0..3 of the tile in flight
    auto issue_rows = [&](const int4& va, const int4& vb, bool hit) {  // a miss reads row 0 with weight 0 (no branches)
      const int idv[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float* row = lv_feat + (size_t)(hit ? (unsigned int)idv[c] : 0u) * F;
#if SHINE_V5_ABL & 8
        ra0[c] = make_float4((float)idv[c], 1.f, 2.f, 3.f);
        ra1[c] = ra0[c];
        (void)row;
#else
        ra0[c] = *reinterpret_cast<const float4*>(row);
        ra1[c] = *reinterpret_cast<const float4*>(row + 4);
#endif
      }
#if !(SHINE_V5_ABL & 8)
#pragma unroll
      for (int c = 4; c < 8; ++c) {  // lane l's 16 bytes land at base + 16 l
        const float* row = lv_feat + (size_t)(hit ? (unsigned int)idv[c] : 0u) * F;
        __builtin_amdgcn_global_load_lds(to_global(row), to_lds(s_dma[pl] + (2 * (c - 4)) * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(to_global(row + 4), to_lds(s_dma[pl] + (2 * (c - 4) + 1) * 256), 16, 0, 0);
      }
#endif
    };
    // prologue: the dependent stages of the first tiles, one after the other (once per pipeline)
    int pm2;
    PD cur, nxt;
    int4 cia, cib;
    {
      const int pm0 = load_pm(pos(0)), pm1 = load_pm(pos(1));
      pm2 = load_pm(pos(2));
      cur = load_pd(pos(0), pm0), nxt = load_pd(pos(1), pm1);
      load_ids(cur.slot, cia, cib);
      issue_rows(cia, cib, cur.p >= 0 && cur.slot >= 0);
    }
    int jk = 0;
    int j = 0;
    for (;; ++j) {
      const int base = tile_pos(j);
      if (base < 0) break;
      if (V5_CH > 0 && j % CHD == 0) last_slot = -2;  // a new chunk: no node carries over
      float* const slot = ring + jk * SLOT;
      int* const st_ids = reinterpret_cast<int*>(slot) + S5_IDS + (8 * g) * V3_WP + pt;
      float* const st_w = slot + S5_W + (8 * g) * V3_WP + pt;
      const bool valid = cur.p >= 0;
      const bool hit = valid && cur.slot >= 0;
      const int ids[8] = {cia.x, cia.y, cia.z, cia.w, cib.x, cib.y, cib.z, cib.w};
      // ---- issue: corner ids of j + 1, point data of j + 2, sample index of j + 3 (the rows of j are already in flight)
      int4 nia, nib;
      load_ids(nxt.slot, nia, nib);
      const PD nx2 = load_pd(pos(j + 2), pm2);
      const int pm3 = load_pm(pos(j + 3));
      // ---- node-run boundaries of the ordered stream, all levels at once: bit 16 g + pt of one 64-bit ballot
      const int myslot = hit ? cur.slot : -1;
      const unsigned int validmask = (unsigned int)__ballot(valid) & 0xFFFFu;
      const int prev = row_prev(myslot, last_slot);
      const bool chg = valid && lvl_on && myslot != prev;
      const unsigned long long chg64 = __ballot(chg);
      const unsigned long long hit64 = __ballot(hit);
      last_slot = row_last(myslot);
      // ---- smooth-step weights of this level, in the reference's association (model/feature_octree.py:186-193)
      float w[8];
      Axis X = axis_weight_rt(poly, cur.x0, lv_res), Y = axis_weight_rt(poly, cur.x1, lv_res), Z = axis_weight_rt(poly, cur.x2, lv_res);
      if (EIK && !hit) X.dt = Y.dt = Z.dt = 0.f;  // a miss: every d w_c / d x carries exactly one of these factors
      corner_weights(X.t, Y.t, Z.t, w);
      if (!hit) {
#pragma unroll
        for (int c = 0; c < 8; ++c) w[c] = 0.f;  // padding lanes and misses contribute nothing to f or to the scatter
      }
      // ---- the slot of tile j - K must have been consumed by the scatter wave
      wait_ge(sync + SY_SDONE, j - K + 1, sync + SY_ERR, waited);
      {
        const int mneg = hit ? 0 : -1;  // a miss stages -1 (trash row), never the speculative ids
#pragma unroll
        for (int c = 0; c < 8; ++c) st_ids[c * V3_WP] = ids[c] | mneg;
        if (!EIK) {
#pragma unroll
          for (int c = 0; c < 8; ++c) st_w[c * V3_WP] = w[c];
        } else {  // the decoder wave stages the weights once delta and q are known; it needs the axis factors
          slot[S5_AX + 0 * 64 + lane] = X.t, slot[S5_AX + 1 * 64 + lane] = Y.t, slot[S5_AX + 2 * 64 + lane] = Z.t;
          slot[S5_AX + 3 * 64 + lane] = X.dt, slot[S5_AX + 4 * 64 + lane] = Y.dt, slot[S5_AX + 5 * 64 + lane] = Z.dt;
        }
      }
      // ---- weighted sum of the eight corner rows (and of their x-derivatives in the eikonal build)
      float pf[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) pf[q] = 0.f;
      float Ag[EIK ? 8 : 1][3];
#pragma unroll
      for (int q = 0; q < (EIK ? 8 : 1); ++q) Ag[q][0] = Ag[q][1] = Ag[q][2] = 0.f;
      auto accumulate = [&](int c, const float4& q0, const float4& q1) {
        const float wc = w[c];
        pf[0] = fmaf(wc, q0.x, pf[0]);
        pf[1] = fmaf(wc, q0.y, pf[1]);
        pf[2] = fmaf(wc, q0.z, pf[2]);
        pf[3] = fmaf(wc, q0.w, pf[3]);
        pf[4] = fmaf(wc, q1.x, pf[4]);
        pf[5] = fmaf(wc, q1.y, pf[5]);
        pf[6] = fmaf(wc, q1.z, pf[6]);
        pf[7] = fmaf(wc, q1.w, pf[7]);
        if (EIK) {
          float dwc[3];
          corner_dw(X, Y, Z, c, dwc);
          const float rr[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
          for (int e = 0; e < 3; ++e) {
#pragma unroll
            for (int q = 0; q < 8; ++q) Ag[q][e] = fmaf(dwc[e], rr[q], Ag[q][e]);  // (zero for a miss: dt = 0 above)
          }
        }
      };
      // everything this wave has in flight was issued at least half an iteration ago; the LDS-DMA rows have no VGPR the
      // compiler could wait on, so the wait is explicit (and total: the counter is in order)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int c = 0; c < 4; ++c) accumulate(c, ra0[c], ra1[c]);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#if SHINE_V5_ABL & 8
        const float4 q0 = make_float4((float)ids[4 + c], 1.f, 2.f, 3.f), q1 = q0;
#else
        const float4 q0 = *reinterpret_cast<const float4*>(s_dma[pl] + (2 * c) * 256 + 4 * lane);
        const float4 q1 = *reinterpret_cast<const float4*>(s_dma[pl] + (2 * c + 1) * 256 + 4 * lane);
#endif
        accumulate(4 + c, q0, q1);
      }
      // ---- reduce-scatter of the per-level sums over the point's four lanes: lane g ends with features (2g, 2g+1)
      {
        float h4[4], f2[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) h4[q] = xsum32(pf[q], pf[4 + q]);  // g < 2: features q, g >= 2: features 4 + q
#pragma unroll
        for (int t = 0; t < 2; ++t) f2[t] = xsum16(h4[t], h4[2 + t]);  // even g: t, odd g: 2 + t
        *reinterpret_cast<float2*>(slot + S5_F + 2 * lane) = make_float2(f2[0], f2[1]);
      }
      if (EIK) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          float h4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) h4[q] = xsum32(Ag[q][e], Ag[4 + q][e]);
#pragma unroll
          for (int t = 0; t < 2; ++t) slot[S5_A2 + (3 * t + e) * 64 + lane] = xsum16(h4[t], h4[2 + t]);
        }
      }
      // ---- issue the rows of tile j + 1: its corner ids landed with the wait above, the LDS landing zone has been read (the
      // features are reduced and staged, so nothing of tile j but its point data is live any more)
      __builtin_amdgcn_sched_barrier(0);
      wave_lds_fence();  // (the reads of the landing zone are done before the next DMA may overwrite it)
      if (tile_pos(j + 1) >= 0) issue_rows(nia, nib, nxt.p >= 0 && nxt.slot >= 0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- meta: label / weight / output position of the 16 points, the three masks, the per-level hit bits
      {
        const int po = a.pool_mode ? base + pt : cur.p;  // where this point's outputs go
        if (g == 0) {
          slot[S5_META + pt] = cur.label;
          slot[S5_META + 16 + pt] = cur.weight;
          reinterpret_cast<int*>(slot)[S5_META + 32 + pt] = valid ? po : -1;
        } else if (g == 1) {
          const int k = pt;  // words 48..: chg lo, chg hi, hit lo, hit hi, validmask, -, -, -, hit bits of levels 0..3
          int v = 0;
          if (k == 0) v = (int)(unsigned int)chg64;
          if (k == 1) v = (int)(unsigned int)(chg64 >> 32);
          if (k == 2) v = (int)(unsigned int)hit64;
          if (k == 3) v = (int)(unsigned int)(hit64 >> 32);
          if (k == 4) v = (int)validmask;
          if (k >= 8 && k < 12) v = (int)((hit64 >> (16 * (k - 8))) & 0xFFFFull);
          reinterpret_cast<int*>(slot)[S5_META + 48 + k] = v;
        }
      }
      publish(sync + SY_GPUB, j + 1);
      // ---- rotate the pipeline registers
      cur = nxt, nxt = nx2, cia = nia, cib = nib, pm2 = pm3;
      if (++jk == K) jk = 0;
      if (SHINE_V5_DYN && (j + 1) % CHD == 0) {  // on to the chunk claimed earlier; claim the one after it
        ++kcur;
        gc_cur = gc_next;
        gc_next = claim();
      }
    }
    tiles_done = j;
    if (SHINE_V5_DYN) *lds_word(sync + SY_TOTAL) = j;  // (after the last tile's publish: LDS writes of a wave are in order)
  } else {
    // ============================================================================================ S: scatter wave
    int run_id[LCAP], run_hit[LCAP];
    float run_acc[LCAP];
#pragma unroll
    for (int s = 0; s < LCAP; ++s) {
      run_id[s] = -1;
      run_acc[s] = 0.f;
      run_hit[s] = 0;
    }
    if (SHINE_V5_PRIO) __builtin_amdgcn_s_setprio(3);
    int lane_o = lane;
    int jk = 0;
    for (int j = 0;; ++j) {
      if (!wait_tile(sync + SY_DPUB0 + (ND == 2 ? (j & 1) : 0), sync + SY_TOTAL, j, sync + SY_ERR, waited)) break;
      ++tiles_done;
      asm volatile("" : "+v"(lane_o));
      float* const slot = ring + jk * SLOT;
      const int* const sc_ids = reinterpret_cast<const int*>(slot) + S5_IDS + (lane_o >> 3) * V3_WP;  // + s * 8 * V3_WP + point
      const float* const sc_w = slot + S5_W + (lane_o >> 3) * V3_WP;
      const float* const sc_df = slot + S5_DF + (lane_o & 7) * V3_DFP;
      // the tile's masks (wave-uniform): one read, five lane broadcasts
      unsigned long long chg64, hit64;
      {
        const int mv = reinterpret_cast<const int*>(slot)[S5_META + 48 + (lane_o & 7)];
        const unsigned int c0 = (unsigned int)__builtin_amdgcn_readlane(mv, 0), c1 = (unsigned int)__builtin_amdgcn_readlane(mv, 1);
        const unsigned int h0 = (unsigned int)__builtin_amdgcn_readlane(mv, 2), h1 = (unsigned int)__builtin_amdgcn_readlane(mv, 3);
        chg64 = ((unsigned long long)c1 << 32) | c0;
        hit64 = ((unsigned long long)h1 << 32) | h0;
      }
      float dfr[V3_TP];
#pragma unroll
      for (int jj = 0; jj < V3_TP / 4; ++jj) {
        const float4 v = *reinterpret_cast<const float4*>(sc_df + 4 * jj);
        dfr[4 * jj] = v.x, dfr[4 * jj + 1] = v.y, dfr[4 * jj + 2] = v.z, dfr[4 * jj + 3] = v.w;
      }
      // staged operands of one level: weights and ids of this lane's corner for the 16 points (8 x 16-B LDS reads); level
      // s + 1's are requested BEFORE level s is walked (the walk is a chain of scalar branches)
      float4 wq[2][V3_TP / 4];
      int4 iq[2][V3_TP / 4];
#pragma unroll
      for (int jj = 0; jj < V3_TP / 4; ++jj) {
        wq[0][jj] = *reinterpret_cast<const float4*>(sc_w + 4 * jj);
        iq[0][jj] = *reinterpret_cast<const int4*>(sc_ids + 4 * jj);
      }
#pragma unroll
      for (int s = 0; s < L; ++s) {
        if (s + 1 < L) {
#pragma unroll
          for (int jj = 0; jj < V3_TP / 4; ++jj) {
            wq[(s + 1) & 1][jj] = *reinterpret_cast<const float4*>(sc_w + ((s + 1) * 8) * V3_WP + 4 * jj);
            iq[(s + 1) & 1][jj] = *reinterpret_cast<const int4*>(sc_ids + ((s + 1) * 8) * V3_WP + 4 * jj);
          }
        } else {
          // the last level's operands are in registers: the slot can go back to the gather wave while it is walked
          publish(sync + SY_SDONE, j + 1);
        }
        float* gbase = a.lv[s].grad;
        if (gbase && !(SHINE_V5_ABL & 4)) {
          float wr[V3_TP];
          int idr[V3_TP];
#pragma unroll
          for (int jj = 0; jj < V3_TP / 4; ++jj) {
            const float4 v = wq[s & 1][jj];
            const int4 u = iq[s & 1][jj];
            wr[4 * jj] = v.x, wr[4 * jj + 1] = v.y, wr[4 * jj + 2] = v.z, wr[4 * jj + 3] = v.w;
            idr[4 * jj] = u.x, idr[4 * jj + 1] = u.y, idr[4 * jj + 2] = u.z, idr[4 * jj + 3] = u.w;
          }
          int rid = run_id[s], rhit = run_hit[s];
          float racc = run_acc[s];
          const unsigned int cm = (unsigned int)(chg64 >> (16 * s)) & 0xFFFFu;
          const unsigned int hm = (unsigned int)(hit64 >> (16 * s)) & 0xFFFFu;
          unsigned char* const tb = a.touched[s];
#if SHINE_V5_PREFIX
          {
            f32x16 wv, dv;
            i32x16 iv;
#pragma unroll
            for (int p2 = 0; p2 < V3_TP; ++p2) wv[p2] = wr[p2], dv[p2] = dfr[p2], iv[p2] = idr[p2];
            scatter_level_prefix<!(SHINE_V5_ABL & 1)>(wv, iv, dv, cm, hm, sq, gbase, tb, rid, rhit, racc);
          }
#else
          if (cm == 0u) {  // no node boundary inside this tile at this level (the usual case at the coarse levels)
#pragma unroll
            for (int p2 = 0; p2 < V3_TP; ++p2) racc = fmaf(wr[p2], dfr[p2], racc);
          } else {
#pragma unroll
            for (int p2 = 0; p2 < V3_TP; ++p2) {
              if (cm & (1u << p2)) {  // a new node (or a run of misses) starts here: close the open run
                if (rhit && !(SHINE_V5_ABL & 1)) atomic_add_f32(gbase + (unsigned int)rid, racc);  // scalar branch
                racc = 0.f;
                rid = (idr[p2] << 3) | sq;  // float offset of this lane's (corner row, feature)
                rhit = (int)((hm >> p2) & 1u);
                // touched-row flags (unique(hierarchical_indices) without -1, for shine_regularize / the active-row
                // optimiser): set at the run start of every hit node by one lane per corner
                if (tb && rhit && sq == 0) tb[idr[p2]] = 1;
              }
              racc = fmaf(wr[p2], dfr[p2], racc);  // misses and padding lanes staged w = 0
            }
          }
#endif
          run_id[s] = rid;
          run_hit[s] = rhit;
          run_acc[s] = racc;
        }
      }
      if (++jk == K) jk = 0;
    }
    // ---- end of the pipeline's range: flush the open node runs
#pragma unroll
    for (int s = 0; s < L; ++s) {
      float* gbase = a.lv[s].grad;
      if (gbase && run_hit[s]) atomic_add_f32(gbase + (unsigned int)run_id[s], run_acc[s]);
    }
  }

  // ------------------------------------------------------------------------------------------------ flush
  if (SHINE_V5_PROF && a.prof && lane == 0) {  // [role, tiles, setup, loop, of which polling, -, -, -] per wave
    const long long now = clk();
    long long* o = a.prof + ((long long)blockIdx.x * (V5_NT / 64) + wv) * 8;
    o[0] = role, o[1] = tiles_done * (role < ND ? ND : 1), o[2] = t_loop - t_start, o[3] = now - t_loop, o[4] = waited, o[5] = o[6] = o[7] = 0;
  }
  __syncthreads();  // every role is done with the ring: it now holds the partial vectors (8 decoder + 4 scatter waves)
  if (role < ND) {
    float* wvec = s_ring + (ND * pl + role) * DVEC;
#pragma unroll
    for (int s = 0; s < L; ++s)
#pragma unroll
      for (int t = 0; t < 2; ++t) {  // features 2g + t of the 16 points of the DPP row
        const float v = row16_sum(trash[s][t]);
        if (pt == 0) wvec[SHINE_MLP_PARAMS + s * 8 + 2 * g + t] = v;
      }
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
          const float w3v = row16_sum(dw3c[4 * m + r]);  // channel 16 m + 4 g + r over the 16 points of the DPP row
          if (pt == 0) wvec[MLP_W3 + 16 * m + 4 * g + r] = w3v;
          if (EIK) {
            const float b2v = row16_sum(db2c[4 * m + r]), b1v = row16_sum(db1c[4 * m + r]);
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
    const double ls = wave_sum_d((double)loss_acc), cs = wave_sum_d((double)cnt_acc);
    const double es = EIK ? wave_sum_d((double)eik_acc) : 0.0;
    if (lane == 0) {
      atomicAdd(&s_loss[0], ls);
      atomicAdd(&s_loss[1], cs);
      if (EIK) atomicAdd(&s_loss[2], es);
    }
  }
  __syncthreads();
  const int mlp_lo = a.decoder_grad_on ? 0 : SHINE_MLP_PARAMS;  // a frozen decoder has no sums to move
  float* dst = a.partials + (long long)blockIdx.x * PART_STRIDE;
  for (int idx = tid; idx < PART_TRASH + L * 8; idx += V5_NT) {
    float v = 0.f;
    if (idx >= mlp_lo) {
      if (idx < PART_TRASH) {
#pragma unroll
        for (int w = 0; w < N_DVEC; ++w) v += s_ring[w * DVEC + idx];
      } else {  // PART_TRASH == SHINE_MLP_PARAMS: the trash sums sit right behind the decoder grads in a wave's vector
#pragma unroll
        for (int w = 0; w < N_DVEC; ++w) v += s_ring[w * DVEC + idx];
      }
    }
    dst[idx] = v;
  }
  for (int idx = PART_TRASH + L * 8 + tid; idx < PART_FLOATS; idx += V5_NT) dst[idx] = 0.f;
  if (tid == 0) {
    int bad = 0;
#pragma unroll
    for (int q = 0; q < V5_PIPES; ++q) bad |= s_sync[q][SY_ERR];
    double* dl = reinterpret_cast<double*>(dst + PART_LOSS);
    dl[0] = bad ? __builtin_nan("") : s_loss[0];  // a timed-out hand-off must not pass for a result
    dl[1] = s_loss[1];
    dl[2] = s_loss[2];
  }
}

constexpr int v5_nd(bool eik) { return eik ? 1 : 2; }

V2Geometry v5_geometry(long long n, bool eik) {
  V2Geometry g;
  long long tiles = (n + V3_TP - 1) / V3_TP;
  if (tiles < 1) tiles = 1;
  g.tiles = tiles;
  g.wg_waves = (v5_nd(eik) + 2) * V5_PIPES;
  const long long pipes = (tiles + 3) / 4;  // at least ~4 tiles per pipeline, one workgroup per CU at most
  long long blocks = (pipes + V5_PIPES - 1) / V5_PIPES;
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  g.blocks = blocks;
  g.waves = blocks * g.wg_waves;
  g.chunk = 0;
  return g;
}

long long v5_lds_bytes(bool eik) {
  const long long K = eik ? 2 : 3, slot = eik ? S5_EIK_FLOATS : S5_BCE_FLOATS;
  return (long long)sizeof(float) * (V3_OPTOTAL + 100 + V5_PIPES * K * slot + V5_PIPES * v5_nd(eik) * V3_R2 +
                                     V5_PIPES * 2048) + 4 * sizeof(double) +
         V5_PIPES * SY_WORDS * sizeof(int);
}

template <int L, bool EIK>
static void launch_v5(const V1Args& a, const V2Geometry& g, hipStream_t st) {
  constexpr int ND = v5_nd(EIK);
  hipLaunchKernelGGL((k_step_v5<L, EIK, ND>), dim3((unsigned)g.blocks), dim3((ND + 2) * V5_PIPES * 64), 0, st, a);
}

}  // namespace shine

using namespace shine;

// same contract as shine_train_step_v3 (shine_step_v3.hip); planned / pool batches, needs the workspace
extern "C" int shine_train_step_v5(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                                   const float* sdf_label, const float* weight, const int32_t* perm,
                                   const int32_t* slots, const int64_t* n_surf, int64_t n, const float* const* feats,
                                   const int64_t* rows, const float* const* mlp, float* pred_out, float* grad_x_out,
                                   float* const* grad_feats, float* const* grad_mlp, double* loss_parts,
                                   unsigned char* const* touched, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  if (!slots) return set_error(SHINE_E_INVALID, "shine_train_step_v5: needs a planned batch (slots)");
  V1Args a = {};
  int rc = fill_step_args(&a, t, cfg, coord, sdf_label, weight, perm, slots, n_surf, n, feats, rows, mlp, pred_out,
                          grad_x_out, grad_feats, grad_mlp, loss_parts, touched);
  if (rc != SHINE_OK) return rc;
  if (n == 0) return SHINE_OK;
  if (n >= (1ll << 31)) return set_error(SHINE_E_INVALID, "shine_train_step_v5: more than 2^31 points per launch");
  const V2Geometry g = v5_geometry(n, cfg->eikonal_on != 0);
  a.tiles = g.tiles;
  a.waves_total = g.waves;
  const size_t need = (size_t)g.blocks * PART_STRIDE * sizeof(float);
  if (!workspace || workspace_bytes < need)
    return set_error(SHINE_E_INVALID, "shine_train_step_v5: workspace too small (shine_train_step_workspace_bytes)");
  a.partials = (float*)workspace;
  a.prof = g_prof_buffer;
  hipStream_t st = (hipStream_t)stream;
  if (cfg->eikonal_on) {
    switch (cfg->n_levels) {
      case 1: launch_v5<1, true>(a, g, st); break;
      case 2: launch_v5<2, true>(a, g, st); break;
      case 3: launch_v5<3, true>(a, g, st); break;
      default: launch_v5<4, true>(a, g, st); break;
    }
  } else {
    switch (cfg->n_levels) {
      case 1: launch_v5<1, false>(a, g, st); break;
      case 2: launch_v5<2, false>(a, g, st); break;
      case 3: launch_v5<3, false>(a, g, st); break;
      default: launch_v5<4, false>(a, g, st); break;
    }
  }
  SHINE_HIP_CHECK(hipGetLastError());
  if (!(a.ablate & 32)) {  // (ablate bit 32: measurement only — time the dominant kernel by itself)
    hipLaunchKernelGGL(k_reduce_partials, dim3((PART_FLOATS + 63) / 64), dim3(1024), 0, st, a, (int)g.blocks, Pass1Args{});
    SHINE_HIP_CHECK(hipGetLastError());
  }
  return SHINE_OK;
}
