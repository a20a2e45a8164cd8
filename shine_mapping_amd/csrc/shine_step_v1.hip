// shine_step_v1.hip — the fused SHINE training step for gfx950 (BCE path), MI355X-first.
//
//   query    FeatureOctree.query_feature   model/feature_octree.py:199-244
//   decode   Decoder.sdf                   model/decoder.py:49-63
//   loss     sdf_bce_loss                  utils/loss.py:17-24
//   backward cur_loss.backward()           shine_batch.py:208-209 (closed form)
//
// One wave owns a contiguous run of the Morton-sorted batch and walks it in tiles of 32 points.
//   lane = (pt = lane & 31, h = lane >> 5): the two half-waves hold features 4h..4h+3 of the same 32 points,
//   which is exactly the B-operand / C-accumulator shape of v_mfma_f32_32x32x2_f32 (exact fp32):
//     D[32 channels x 32 points] += A[32 x 2] . B[2 x 32],   lane l: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
//     acc reg r of lane l = D[row (r&3) + 8(r>>2) + 4(l>>5)][col l&31].
//   * decoder forward/backward are chained MFMAs that never leave registers: the k-order of each product is
//     permuted to the accumulator row order of the previous one (k(t,h) = rowidx(t,h)), so the ReLU'd
//     accumulator register t IS the next B operand; the matching A operands are pre-permuted once per block
//     into LDS (s_opA);
//   * the weight-grad GEMMs contract over POINTS, so their operands are the transposes: d2/h1/d1/f go once
//     through a padded [32][33] LDS tile per wave, and accumulate into MFMA accumulators that live in registers
//     for the whole kernel (flushed once per wave);
//   * feature grads: lane = (corner c = lane>>3, feature q = lane&7); the wave walks its sorted points and
//     keeps a running sum per level while the node (hence the 8 corner rows) stays the same, so one
//     64-lane global_atomic_add_f32 (8 rows x 32 B) is issued per NODE RUN instead of per point; misses
//     (index -1, the trash row :205,231) are summed in registers for the whole kernel;
//   * decoder / trash-row / loss sums leave the block as one partial vector in the caller's workspace and a
//     second tiny kernel adds them up: no hot-spot atomics, deterministic.
#include "shine_internal.hpp"

namespace shine {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LCAP = 4;        // featured levels handled by this kernel (tree_level_feat <= 4 in every yaml)
constexpr int TP = 33;         // transpose tile pitch (floats)
constexpr int WP = 36;         // pitch of the [corner][point] / [feature][point] staging rows (conflict-free b128 reads)
constexpr int U_IDS = LCAP * 8 * WP;             // ids  [LCAP][8 corners][WP]      int32 (point-contiguous)
constexpr int U_W = LCAP * 8 * WP;               // w    [LCAP][8 corners][WP]      (point-contiguous)
constexpr int U_FLOATS = U_IDS + U_W;            // 2176 >= 2*32*TP = 2112 (the transpose tiles alias this region)
constexpr int DF_FLOATS = 8 * WP;                // df   [8 features][WP]
constexpr int WAVE_FLOATS = U_FLOATS + DF_FLOATS;
constexpr int OP_A1 = 0, OP_A2 = 4 * 64, OP_A2T = 20 * 64, OP_A1T = 36 * 64, OP_TOTAL = 52 * 64;
constexpr int SB_B1 = 0, SB_B2 = 32, SB_W3 = 64, SB_B3 = 96;
constexpr int PART_TRASH = SHINE_MLP_PARAMS;               // + s*8 + q
constexpr int PART_FLOATS = PART_TRASH + SHINE_MAX_LEVELS * 8;  // 1441
constexpr int PART_LOSS = 1444;                            // float index of double[2] {loss, count} (8-B aligned)
constexpr int PART_STRIDE = 1448;

static_assert(LCAP % 2 == 0, "levels are processed in pairs");
static_assert(2 * 32 * TP <= U_FLOATS, "staging region must hold the two transpose tiles");
static_assert((PART_LOSS * 4) % 8 == 0 && PART_LOSS >= PART_FLOATS, "loss slot");

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
  const float* coord;
  const float* label;
  const int* perm;
  const float* mlp[6];
  float* pred;
  float* grad_mlp[6];
  double* loss_parts;
  float* partials;
  float* feat_rw[LCAP];  // same tables, writable: the trash row is re-zeroed in-kernel (set_zero, :78-81)
  long long n;
  long long chunk;
  int n_levels;
  int reduction_sum;
  int decoder_grad_on;
  int poly;
  int ablate;  // debug only (kernel_variant >> 8): 1 no feature atomics, 2 no weight-grad phase, 4 no scatter phase,
               // 8 no row gathers, 16 no probe (every point misses)
  float sigma;
  float inv_n;
  long long* prof;  // debug: per-wave phase cycle counters [waves][8] (shine_debug_set_profile_buffer) or null
};

__device__ __forceinline__ long long clk() { return (long long)__builtin_readcyclecounter(); }

__device__ __forceinline__ int rowidx(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// wave-local LDS hand-off: DS ops of one wave execute in order; this only stops the compiler reordering them
// (a workgroup-scope __builtin_amdgcn_fence would also drain vmcnt, i.e. wait for every gather/atomic in flight)
__device__ __forceinline__ void wave_lds_fence() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// sum over the 16 lanes of a DPP row, result in every lane of the row (4 VALU ops, no LDS traffic)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

__device__ __forceinline__ Axis axis_weight_rt(bool poly, float x, float res) {
  return poly ? axis_weight<true>(x, res, res * 0.5f) : axis_weight<false>(x, res, res * 0.5f);
}

template <int L, bool PROF>
__global__ __launch_bounds__(256, 2) void k_step_v1(V1Args a) {
  __shared__ float s_opA[OP_TOTAL];
  __shared__ float s_bias[100];
  __shared__ float s_part[PART_STRIDE];
  __shared__ double s_loss[2];
  __shared__ float s_wave[4][WAVE_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int pt = lane & 31, h = lane >> 5;
  const bool poly = a.poly != 0;
  long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long tk = PROF ? clk() : 0;
#define SHINE_STAMP(k)            \
  if (PROF) {                     \
    long long now__ = clk();      \
    pc[k] += now__ - tk;          \
    tk = now__;                   \
  }

  // ---- per-block setup: A operands in MFMA lane order, biases, zeroed partial vector
  for (int idx = tid; idx < OP_TOTAL; idx += 256) {
    const int t = idx >> 6, l = idx & 63, li = l & 31, lh = l >> 5;
    float v;
    if (t < 4) {
      v = a.mlp[0][li * F + 4 * lh + t];                       // W1[ch=li][k = 4h+t]
    } else if (t < 20) {
      v = a.mlp[2][li * H + rowidx(t - 4, lh)];                // W2[out=li][in = rowidx]
    } else if (t < 36) {
      v = a.mlp[2][rowidx(t - 20, lh) * H + li];               // W2^T[in=li][out = rowidx]
    } else {
      v = li < F ? a.mlp[0][rowidx(t - 36, lh) * F + li] : 0.f;  // W1^T[feat=li][ch = rowidx], rows >= 8 zero
    }
    s_opA[idx] = v;
  }
  for (int idx = tid; idx < 32; idx += 256) {
    s_bias[SB_B1 + idx] = a.mlp[1][idx];
    s_bias[SB_B2 + idx] = a.mlp[3][idx];
    s_bias[SB_W3 + idx] = a.mlp[4][idx];
  }
  if (tid == 0) {
    s_bias[SB_B3] = a.mlp[5][0];
    s_loss[0] = 0.0;
    s_loss[1] = 0.0;
  }
  for (int idx = tid; idx < PART_STRIDE; idx += 256) s_part[idx] = 0.f;
  __syncthreads();

  float* U = s_wave[wv];
  int* U_ids = reinterpret_cast<int*>(U);        // [LCAP][8][WP]
  float* U_w = U + U_IDS;                        // [LCAP][8][WP]
  float* TL = U;                                 // [32][TP]   (aliases the staging above, used after the scatter)
  float* TR = U + 32 * TP;                       // [32][TP]
  float* s_df = U + U_FLOATS;                    // [8][WP]

  const float b3 = s_bias[SB_B3];

  f32x16 accW2, accW1;
  float dw3c[16];   // sum_p delta_p h2[ch][p] for this lane's point column, reduced over lanes once per wave
  float db2acc = 0.f;  // sum_p d2[ch = lane&31][p] for the points of this half-wave's parity (transposed reads)
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    accW2[r] = 0.f;
    accW1[r] = 0.f;
    dw3c[r] = 0.f;
  }
  float db3 = 0.f;
  double loss_acc = 0.0, cnt_acc = 0.0;
  int run_id[LCAP];      // corner id (this lane's corner) of the node run in progress, -1: none / a run of misses
  int last_slot[LCAP];   // node (hash slot) of the previous point, wave-uniform: carries runs across tiles
  float run_acc[LCAP], trash_acc[LCAP];
#pragma unroll
  for (int s = 0; s < LCAP; ++s) {
    run_id[s] = -1;
    last_slot[s] = -2;
    run_acc[s] = 0.f;
    trash_acc[s] = 0.f;
  }
  const int sc = lane >> 3, sq = lane & 7;  // scatter role: corner, feature

  const long long wave_g = (long long)blockIdx.x * 4 + wv;
  const long long begin = wave_g * a.chunk;
  const long long end = (begin + a.chunk < a.n) ? begin + a.chunk : a.n;
  SHINE_STAMP(0)  // setup

  // software prefetch of the {perm -> coord, label} chain: tile t+1's point is fetched while tile t computes
  long long np = 0;
  float nx0 = 0.f, nx1 = 0.f, nx2 = 0.f, nlabel = 0.f;
  bool nvalid = begin + pt < end;
  if (nvalid) {
    np = a.perm ? (long long)a.perm[begin + pt] : begin + pt;
    nx0 = a.coord[3 * np];
    nx1 = a.coord[3 * np + 1];
    nx2 = a.coord[3 * np + 2];
    nlabel = a.label[np];
  }

  for (long long base = begin; base < end; base += 32) {
    const bool valid = nvalid;
    const long long p = np;
    const float x0 = nx0, x1 = nx1, x2 = nx2, label = nlabel;
    {
      const long long ni = base + 32 + pt;
      nvalid = ni < end;
      np = 0;
      nx0 = nx1 = nx2 = nlabel = 0.f;
      if (nvalid) {
        np = a.perm ? (long long)a.perm[ni] : ni;
        nx0 = a.coord[3 * np];
        nx1 = a.coord[3 * np + 1];
        nx2 = a.coord[3 * np + 2];
        nlabel = a.label[np];
      }
    }

    // ================================================================ phase 1: query (all levels)
    // Straight-line across levels so the L independent {probe -> ids -> 8 rows} chains overlap: first-slot key
    // loads for every level, then ids, then rows (a miss reads row 0 with weight 0 instead of branching).
    float f4[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned int chgmask[LCAP], hitmask[LCAP];
    const unsigned int validmask = (unsigned int)__ballot(valid);
#pragma unroll
    for (int s = 0; s < LCAP; ++s) {
      chgmask[s] = 0;
      hitmask[s] = 0;
    }
    // leaf-level key once; a parent's Morton code is the child's >> 3 (quantisation is exact power-of-two scaling)
    unsigned long long key[LCAP];
    unsigned int slot0[LCAP];
    unsigned long long k0[LCAP];
    int slot[LCAP];
    {
      const float rl = a.lv[L - 1].res;
      const unsigned long long kleaf = morton3(quantize(x0, rl), quantize(x1, rl), quantize(x2, rl));
#pragma unroll
      for (int s = 0; s < L; ++s) {
        key[s] = kleaf >> (3 * (L - 1 - s));
        slot0[s] = hash_slot(key[s], a.lv[s].shift);
        k0[s] = a.lv[s].keys[slot0[s]];
      }
    }
#pragma unroll
    for (int s = 0; s < L; ++s) {
      const V1Level& Lv = a.lv[s];
      int sl = -1;
      if (k0[s] == key[s]) {
        sl = (int)slot0[s];
      } else if (k0[s] != EMPTY_KEY) {  // collision on the first slot: rare, walk the probe sequence
        unsigned int q = (slot0[s] + 1) & Lv.mask;
        for (unsigned int n = 0; n < Lv.mask; ++n) {
          const unsigned long long kk = Lv.keys[q];
          if (kk == key[s]) {
            sl = (int)q;
            break;
          }
          if (kk == EMPTY_KEY) break;
          q = (q + 1) & Lv.mask;
        }
      }
      slot[s] = (valid && !(a.ablate & 16)) ? sl : -1;
    }
#pragma unroll
    for (int sp = 0; sp < L; sp += 2) {  // ids + rows of two levels in flight at a time (register budget)
    int4 i0[LCAP], i1[LCAP];
#pragma unroll
    for (int s = sp; s < (sp + 2 < L ? sp + 2 : L); ++s) {
      const V1Level& Lv = a.lv[s];
      const int sl = slot[s] >= 0 ? slot[s] : 0;
      i0[s] = Lv.vals[2 * sl];
      i1[s] = Lv.vals[2 * sl + 1];
    }
#pragma unroll
    for (int s = sp; s < (sp + 2 < L ? sp + 2 : L); ++s) {
      {
        const V1Level& Lv = a.lv[s];
        const bool hit = slot[s] >= 0;
        // node-run boundaries of the sorted stream (wave-uniform bit masks over the 32 points of the tile)
        int prev = __shfl_up(slot[s], 1, 64);
        if (pt == 0) prev = last_slot[s];
        chgmask[s] = (unsigned int)__ballot(valid && slot[s] != prev);
        hitmask[s] = (unsigned int)__ballot(hit);
        last_slot[s] = __builtin_amdgcn_readlane(slot[s], 31);
        Axis X = axis_weight_rt(poly, x0, Lv.res), Y = axis_weight_rt(poly, x1, Lv.res),
             Z = axis_weight_rt(poly, x2, Lv.res);
        float w[8];
        corner_weights(X.t, Y.t, Z.t, w);
        if (!valid) {
#pragma unroll
          for (int c = 0; c < 8; ++c) w[c] = 0.f;  // padding lanes contribute nothing anywhere
        }
        // staging for the scatter: ids [pt][8] (h=0 writes corners 0-3, h=1 corners 4-7), w [corner][pt]
        {
          const int m = hit ? 0 : -1;  // a miss stages -1 (trash row), never the speculative ids
          U_ids[(s * 8 + 4 * h + 0) * WP + pt] = (h == 0 ? i0[s].x : i1[s].x) | m;
          U_ids[(s * 8 + 4 * h + 1) * WP + pt] = (h == 0 ? i0[s].y : i1[s].y) | m;
          U_ids[(s * 8 + 4 * h + 2) * WP + pt] = (h == 0 ? i0[s].z : i1[s].z) | m;
          U_ids[(s * 8 + 4 * h + 3) * WP + pt] = (h == 0 ? i0[s].w : i1[s].w) | m;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) U_w[(s * 8 + 4 * h + c) * WP + pt] = h == 0 ? w[c] : w[4 + c];
        if (!(a.ablate & 8)) {
          const int ids[8] = {i0[s].x, i0[s].y, i0[s].z, i0[s].w, i1[s].x, i1[s].y, i1[s].z, i1[s].w};
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float wz = hit ? w[c] : 0.f;
            const long long row = hit ? (long long)ids[c] : 0;
            const float4 r = *reinterpret_cast<const float4*>(Lv.feat + row * F + 4 * h);
            f4[0] = fmaf(wz, r.x, f4[0]);
            f4[1] = fmaf(wz, r.y, f4[1]);
            f4[2] = fmaf(wz, r.z, f4[2]);
            f4[3] = fmaf(wz, r.w, f4[3]);
          }
        }
      }
    }
    }  // level pairs

    SHINE_STAMP(1)  // query
    // ================================================================ phase 2: decoder forward (MFMA chain)
    f32x16 c1, c2;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      c1[r] = s_bias[SB_B1 + rowidx(r, h)];
      c2[r] = s_bias[SB_B2 + rowidx(r, h)];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) c1 = mfma32(s_opA[OP_A1 + t * 64 + lane], f4[t], c1);
    float h1[16], h2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[r] = fmaxf(c1[r], 0.f);
#pragma unroll
    for (int t = 0; t < 16; ++t) c2 = mfma32(s_opA[OP_A2 + t * 64 + lane], h1[t], c2);
    float yp = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      h2[r] = fmaxf(c2[r], 0.f);
      yp = fmaf(s_bias[SB_W3 + rowidx(r, h)], h2[r], yp);
    }
    const float y = yp + __shfl_xor(yp, 32, 64) + b3;
    if (valid && h == 0 && a.pred) a.pred[p] = y;

    SHINE_STAMP(2)  // decoder forward
    // ================================================================ phase 3: loss
    float delta = 0.f;
    if (valid) {
      const float zt = sigmoidf_acc(label / a.sigma);
      if (h == 0) {
        loss_acc += (double)(fmaxf(y, 0.f) - y * zt + log1pf(expf(-fabsf(y))));
        cnt_acc += 1.0;
      }
      delta = (sigmoidf_acc(y) - zt) * a.inv_n;
    }

    // ================================================================ phase 4: backward through the decoder
    float d2[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      d2[r] = h2[r] > 0.f ? delta * s_bias[SB_W3 + rowidx(r, h)] : 0.f;
      dw3c[r] = fmaf(delta, h2[r], dw3c[r]);
    }
    if (h == 0) db3 += delta;
    f32x16 e1, e0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      e1[r] = 0.f;
      e0[r] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) e1 = mfma32(s_opA[OP_A2T + t * 64 + lane], d2[t], e1);
    float d1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      d1[r] = h1[r] > 0.f ? e1[r] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) e0 = mfma32(s_opA[OP_A1T + t * 64 + lane], d1[t], e0);
    // rows 0..7 of e0 = d loss / d f ; lane (pt,h) holds rows 4h..4h+3 in regs 0..3
#pragma unroll
    for (int q = 0; q < 4; ++q) s_df[(4 * h + q) * WP + pt] = e0[q];
    wave_lds_fence();

    SHINE_STAMP(3)  // loss + decoder backward
    // ================================================================ phase 6: feature-grad scatter (run-length)
    // lane = (corner sc, feature sq).  Run boundaries / hits are wave-uniform bit masks, so the loop below has
    // scalar branches only; one 64-lane atomic (8 rows x 32 B) per node run, misses go to a register sum.
    if (!(a.ablate & 4)) {
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {  // 16 points at a time keeps the staged operands in 32 registers
        float dfr[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(s_df + sq * WP + 16 * half + 4 * j);
          dfr[4 * j] = v.x;
          dfr[4 * j + 1] = v.y;
          dfr[4 * j + 2] = v.z;
          dfr[4 * j + 3] = v.w;
        }
#pragma unroll
        for (int s = 0; s < L; ++s) {
          {
            float* gbase = a.lv[s].grad;
            if (gbase) {
              float wr[16];
              int idr[16];
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float4 v = *reinterpret_cast<const float4*>(U_w + (s * 8 + sc) * WP + 16 * half + 4 * j);
                wr[4 * j] = v.x;
                wr[4 * j + 1] = v.y;
                wr[4 * j + 2] = v.z;
                wr[4 * j + 3] = v.w;
                const int4 u = *reinterpret_cast<const int4*>(U_ids + (s * 8 + sc) * WP + 16 * half + 4 * j);
                idr[4 * j] = u.x;
                idr[4 * j + 1] = u.y;
                idr[4 * j + 2] = u.z;
                idr[4 * j + 3] = u.w;
              }
              int rid = run_id[s];
              float racc = run_acc[s], tacc = trash_acc[s];
              const unsigned int cm = (chgmask[s] & validmask) >> (16 * half), hm = hitmask[s] >> (16 * half);
#pragma unroll
              for (int p2 = 0; p2 < 16; ++p2) {
                if (cm & (1u << p2)) {  // a new node (or a run of misses) starts here: close the open run
                  if (rid >= 0 && !(a.ablate & 1)) atomic_add_f32(gbase + (long long)rid * F + sq, racc);
                  racc = 0.f;
                  rid = idr[p2];
                }
                const float v = wr[p2] * dfr[p2];
                if (hm & (1u << p2))
                  racc += v;
                else
                  tacc += v;  // padding lanes carry w = 0
              }
              run_id[s] = rid;
              run_acc[s] = racc;
              trash_acc[s] = tacc;
            }
          }
        }
      }
    }
    wave_lds_fence();

    SHINE_STAMP(4)  // scatter
    // ================================================================ phase 5: decoder weight grads (transposed MFMA)
    if (a.decoder_grad_on && !(a.ablate & 2)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        TL[rowidx(r, h) * TP + pt] = d2[r];
        TR[rowidx(r, h) * TP + pt] = h1[r];
      }
      wave_lds_fence();
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int k = 2 * t + h;
        const float d2t = TL[pt * TP + k];
        db2acc += d2t;                                   // db2[out = pt] rides on the operand that is loaded anyway
        accW2 = mfma32(d2t, TR[pt * TP + k], accW2);     // dW2[out][in] += d2[out][k] * h1[in][k]
      }
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 16; ++r) TL[rowidx(r, h) * TP + pt] = d1[r];
#pragma unroll
      for (int q = 0; q < 4; ++q) TR[(4 * h + q) * TP + pt] = f4[q];
      wave_lds_fence();
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int k = 2 * t + h;
        // B columns 0..7 = f, column 8 = ones: accW1[:,8] accumulates db1 = sum_k d1[ch][k] in the spare MFMA lanes
        const float b = pt < F ? TR[pt * TP + k] : (pt == F ? 1.f : 0.f);
        accW1 = mfma32(TL[pt * TP + k], b, accW1);  // dW1[ch][feat] += d1[ch][k] * f[feat][k]
      }
      wave_lds_fence();
    }
    SHINE_STAMP(5)  // weight grads
  }

  // ---- end of the wave's run: flush the open node runs
#pragma unroll
  for (int s = 0; s < L; ++s) {
    {
      float* gbase = a.lv[s].grad;
      if (gbase && run_id[s] >= 0) atomic_add_f32(gbase + (long long)run_id[s] * F + sq, run_acc[s]);
      // trash row: sum the 8 corner lanes of each feature
      float tsum = trash_acc[s];
      tsum += __shfl_xor(tsum, 8, 64);
      tsum += __shfl_xor(tsum, 16, 64);
      tsum += __shfl_xor(tsum, 32, 64);
      if (sc == 0 && tsum != 0.f) atomicAdd(&s_part[PART_TRASH + s * 8 + sq], tsum);
    }
  }
  if (a.decoder_grad_on) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rowidx(r, h);
      atomicAdd(&s_part[MLP_W2 + row * H + pt], accW2[r]);
      if (pt < F) atomicAdd(&s_part[MLP_W1 + row * F + pt], accW1[r]);
      if (pt == F) atomicAdd(&s_part[MLP_B1 + row], accW1[r]);
      const float w3v = row16_sum(dw3c[r]);
      if ((lane & 15) == 0) atomicAdd(&s_part[MLP_W3 + row], w3v);  // two DPP rows per half-wave
    }
    atomicAdd(&s_part[MLP_B2 + pt], db2acc);  // the two half-waves hold the two point parities
    float b3v = wave_sum(db3);
    if (lane == 0) atomicAdd(&s_part[MLP_B3], b3v);
  }
  {
    double ls = wave_sum_d(loss_acc), cs = wave_sum_d(cnt_acc);
    if (lane == 0) {
      atomicAdd(&s_loss[0], ls);
      atomicAdd(&s_loss[1], cs);
    }
  }
  SHINE_STAMP(6)  // flush
  __syncthreads();
  SHINE_STAMP(7)  // wait for the block
  if (PROF && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) a.prof[wave_g * 8 + k] = pc[k];
  }

  if (a.partials) {
    float* dst = a.partials + (long long)blockIdx.x * PART_STRIDE;
    for (int idx = tid; idx < PART_FLOATS; idx += 256) dst[idx] = s_part[idx];
    if (tid == 0) {
      double* dl = reinterpret_cast<double*>(dst + PART_LOSS);
      dl[0] = s_loss[0];
      dl[1] = s_loss[1];
    }
  } else {
    // no workspace: hot-spot atomics (slower, not deterministic)
    if (a.decoder_grad_on) {
      for (int idx = tid; idx < SHINE_MLP_PARAMS; idx += 256) {
        float v = s_part[idx];
        float* d;
        if (idx < MLP_B1) d = a.grad_mlp[0] + idx;
        else if (idx < MLP_W2) d = a.grad_mlp[1] + (idx - MLP_B1);
        else if (idx < MLP_B2) d = a.grad_mlp[2] + (idx - MLP_W2);
        else if (idx < MLP_W3) d = a.grad_mlp[3] + (idx - MLP_B2);
        else if (idx < MLP_B3) d = a.grad_mlp[4] + (idx - MLP_W3);
        else d = a.grad_mlp[5];
        if (v != 0.f) atomic_add_f32(d, v);
      }
    }
    for (int idx = tid; idx < L * 8; idx += 256) {
      const int s = idx >> 3, q = idx & 7;
      float v = s_part[PART_TRASH + idx];
      if (v != 0.f && a.lv[s].grad) atomic_add_f32(a.lv[s].grad + a.rows[s] * F + q, v);
    }
    if (tid == 0 && a.loss_parts) {
      const double bce = a.reduction_sum ? s_loss[0] : s_loss[0] * (double)a.inv_n;
      atomicAdd(a.loss_parts + 0, bce);
      atomicAdd(a.loss_parts + 2, s_loss[1]);
      atomicAdd(a.loss_parts + 3, bce);
    }
  }
}

// second stage: add the per-block partial vectors into the gradient tensors / loss.
// One 1024-thread block per 64 entries: lane = entry (coalesced 256-B rows), the 16 waves split the blocks.
__global__ __launch_bounds__(1024) void k_reduce_partials(V1Args a, int nblocks) {
  __shared__ float s_red[16][64];
  __shared__ double s_dred[16][2];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + lane;
  const int L = a.n_levels;
  float s = 0.f;
  if (idx < PART_FLOATS) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = part;
    for (; b + 48 < nblocks; b += 64) {
      s0 += a.partials[(long long)b * PART_STRIDE + idx];
      s1 += a.partials[(long long)(b + 16) * PART_STRIDE + idx];
      s2 += a.partials[(long long)(b + 32) * PART_STRIDE + idx];
      s3 += a.partials[(long long)(b + 48) * PART_STRIDE + idx];
    }
    for (; b < nblocks; b += 16) s0 += a.partials[(long long)b * PART_STRIDE + idx];
    s = (s0 + s1) + (s2 + s3);
  }
  s_red[part][lane] = s;
  if (blockIdx.x == 0 && lane < 2) {  // loss / count doubles ride along in block 0
    double d = 0.0;
    for (int b = part; b < nblocks; b += 16)
      d += reinterpret_cast<const double*>(a.partials + (long long)b * PART_STRIDE + PART_LOSS)[lane];
    s_dred[part][lane] = d;
  }
  __syncthreads();
  if (part == 0 && idx < PART_FLOATS) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) tot += s_red[k][lane];
    if (idx < SHINE_MLP_PARAMS) {
      if (a.decoder_grad_on) {
        float* d;
        if (idx < MLP_B1) d = a.grad_mlp[0] + idx;
        else if (idx < MLP_W2) d = a.grad_mlp[1] + (idx - MLP_B1);
        else if (idx < MLP_B2) d = a.grad_mlp[2] + (idx - MLP_W2);
        else if (idx < MLP_W3) d = a.grad_mlp[3] + (idx - MLP_B2);
        else if (idx < MLP_B3) d = a.grad_mlp[4] + (idx - MLP_W3);
        else d = a.grad_mlp[5];
        *d += tot;
      }
    } else {
      const int t = idx - PART_TRASH, sl = t >> 3, q = t & 7;
      if (sl < L && a.lv[sl].grad) a.lv[sl].grad[a.rows[sl] * F + q] += tot;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.loss_parts) {
    double ls = 0.0, cs = 0.0;
    for (int k = 0; k < 16; ++k) {
      ls += s_dred[k][0];
      cs += s_dred[k][1];
    }
    const double bce = a.reduction_sum ? ls : ls * (double)a.inv_n;
    a.loss_parts[0] = bce;
    a.loss_parts[1] = 0.0;
    a.loss_parts[2] = cs;
    a.loss_parts[3] = bce;  // total of the fused terms
  }
  // FeatureOctree.set_zero (model/feature_octree.py:78-81): the fused step never reads the trash row (a miss
  // contributes nothing), so re-zeroing it here is equivalent to zeroing it before the query
  if (blockIdx.x == 1 && threadIdx.x < L * F) {
    const int sl = threadIdx.x / F, q = threadIdx.x % F;
    if (a.feat_rw[sl]) a.feat_rw[sl][a.rows[sl] * F + q] = 0.f;
  }
}

// D[32x32] = A[32x2] . B[2x32] through one v_mfma_f32_32x32x2_f32: pins the operand / accumulator lane maps
__global__ void k_selftest_mfma(const float* A, const float* B, float* D) {
  const int lane = threadIdx.x;
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = mfma32(A[(lane & 31) * 2 + (lane >> 5)], B[(lane >> 5) * 32 + (lane & 31)], c);
#pragma unroll
  for (int r = 0; r < 16; ++r) D[rowidx(r, lane >> 5) * 32 + (lane & 31)] = c[r];
}

struct V1Geometry {
  long long waves, chunk, blocks;
};
static V1Geometry v1_geometry(long long n) {
  V1Geometry g;
  long long tiles = (n + 31) / 32;
  long long max_waves = 256 * 2 * 4;  // 2 resident blocks of 4 waves per CU
  g.waves = tiles < max_waves ? (tiles < 1 ? 1 : tiles) : max_waves;
  long long per = (n + g.waves - 1) / g.waves;
  g.chunk = ((per + 31) / 32) * 32;
  g.waves = (n + g.chunk - 1) / g.chunk;
  if (g.waves < 1) g.waves = 1;
  g.blocks = (g.waves + 3) / 4;
  return g;
}

}  // namespace shine

using namespace shine;

extern "C" size_t shine_train_step_workspace_bytes(const shine_step_config* cfg, int64_t n) {
  (void)cfg;
  if (n <= 0) return 0;
  return (size_t)v1_geometry(n).blocks * PART_STRIDE * sizeof(float);
}

static long long* g_prof_buffer = nullptr;
// debug hook (not part of the public ABI): per-wave phase cycle counters, [waves][8] int64, or NULL to disable
extern "C" void shine_debug_set_profile_buffer(long long* p) { g_prof_buffer = p; }

extern "C" int shine_selftest_mfma(const float* a, const float* b, float* d, void* stream) {
  if (!a || !b || !d) return set_error(SHINE_E_INVALID, "shine_selftest_mfma: null argument");
  hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, d);
  SHINE_HIP_CHECK(hipGetLastError());
  return SHINE_OK;
}

extern "C" int shine_train_step_v1(const shine_tables* t, const shine_step_config* cfg, const float* coord,
                                   const float* sdf_label, const float* weight, const int32_t* perm,
                                   const int64_t* n_surf, int64_t n, const float* const* feats, const int64_t* rows,
                                   const float* const* mlp, float* pred_out, float* grad_x_out,
                                   float* const* grad_feats, float* const* grad_mlp, double* loss_parts,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  (void)weight;
  (void)n_surf;
  (void)grad_x_out;
  if (n < 0 || !feats || !rows || !mlp || !grad_feats || (n > 0 && (!coord || !sdf_label)))
    return set_error(SHINE_E_INVALID, "shine_train_step: null argument");
  if (cfg->n_levels > LCAP || cfg->eikonal_on) return set_error(SHINE_E_INVALID, "shine_train_step_v1: unsupported config");
  V1Args a = {};
  LevelSet ls = {};
  int rc = make_level_set(t, cfg, feats, rows, grad_feats, &ls);
  if (rc != SHINE_OK) return rc;
  for (int s = 0; s < cfg->n_levels; ++s) {
    if (!feats[s]) return set_error(SHINE_E_INVALID, "shine_train_step: null feature level");
    a.lv[s].keys = ls.lv[s].keys;
    a.lv[s].vals = ls.lv[s].vals;
    a.lv[s].feat = ls.lv[s].feat;
    a.lv[s].grad = ls.lv[s].grad;
    a.lv[s].shift = ls.lv[s].shift;
    a.lv[s].mask = ls.lv[s].mask;
    a.lv[s].res = ls.lv[s].res;
    a.rows[s] = ls.lv[s].rows;
    a.feat_rw[s] = const_cast<float*>(feats[s]);
  }
  for (int k = 0; k < 6; ++k) {
    if (!mlp[k]) return set_error(SHINE_E_INVALID, "shine_train_step: null decoder parameter");
    a.mlp[k] = mlp[k];
    if (cfg->decoder_grad_on) {
      if (!grad_mlp || !grad_mlp[k]) return set_error(SHINE_E_INVALID, "shine_train_step: null decoder grad");
      a.grad_mlp[k] = grad_mlp[k];
    }
  }
  if (n == 0) return SHINE_OK;
  V1Geometry g = v1_geometry(n);
  a.coord = coord;
  a.label = sdf_label;
  a.perm = perm;
  a.pred = pred_out;
  a.loss_parts = loss_parts;
  a.n = n;
  a.chunk = g.chunk;
  a.n_levels = cfg->n_levels;
  a.reduction_sum = cfg->reduction_sum;
  a.decoder_grad_on = cfg->decoder_grad_on;
  a.poly = cfg->poly_int_on;
  a.ablate = cfg->kernel_variant >> 8;
  a.prof = g_prof_buffer;
  a.sigma = cfg->sigma;
  a.inv_n = (float)cfg->inv_n;
  const size_t need = (size_t)g.blocks * PART_STRIDE * sizeof(float);
  a.partials = (workspace && workspace_bytes >= need) ? (float*)workspace : nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (!a.partials && loss_parts) SHINE_HIP_CHECK(hipMemsetAsync(loss_parts, 0, 4 * sizeof(double), st));
  const dim3 grid((unsigned)g.blocks), block(256);
  if (a.prof && cfg->n_levels == 4) {  // debug build of the same kernel with s_memtime stamps per phase
    hipLaunchKernelGGL((k_step_v1<4, true>), grid, block, 0, st, a);
  } else {
    a.prof = nullptr;
    switch (cfg->n_levels) {  // the level count is a template parameter: straight-line query code, no guards
      case 1: hipLaunchKernelGGL((k_step_v1<1, false>), grid, block, 0, st, a); break;
      case 2: hipLaunchKernelGGL((k_step_v1<2, false>), grid, block, 0, st, a); break;
      case 3: hipLaunchKernelGGL((k_step_v1<3, false>), grid, block, 0, st, a); break;
      default: hipLaunchKernelGGL((k_step_v1<4, false>), grid, block, 0, st, a); break;
    }
  }
  SHINE_HIP_CHECK(hipGetLastError());
  if (a.partials) {
    hipLaunchKernelGGL(k_reduce_partials, dim3((PART_FLOATS + 63) / 64), dim3(1024), 0, st, a, (int)g.blocks);
    SHINE_HIP_CHECK(hipGetLastError());
  }
  return SHINE_OK;
}
