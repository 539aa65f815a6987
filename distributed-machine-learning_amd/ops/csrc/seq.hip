// Packed training of small sequence-model trials: every resident trial's encoder step in shared launches (gfx950).
//
// The model family is the reference's sample regressor (/root/reference/ray-tune-hpo-regression-sample.py:60-83):
// input projection, L post-norm encoder layers (fused QKV projection, softmax attention, output projection, residual +
// LayerNorm, ReLU feed-forward of width 4d, residual + LayerNorm, dropout at the four usual sites), a linear head on
// the last token, trained in f32 with Adam/AdamW on a squared-error loss (same file, lines 104-119).  One such trial at
// batch 32 x 50 tokens is ~55 kernels of a few workgroups each -- far too small for the chip -- and the reference runs
// them one trial per GPU.  Here a pack of trials with DIFFERENT widths, head counts, depths, dropout rates and
// learning rates shares each launch:
//
//   * every dense layer of layer index l of all slots is ONE grouped exact-f32 MFMA GEMM (gemm_f32.hip): descriptors
//     per slot, an XCD-aware tile list, bias / ReLU / dropout fused in the epilogue; in the backward pass the chain of
//     input gradients is one launch per dense layer (ReLU + dropout mask, or the residual gradient added, in the
//     epilogue), and the weight gradients of ALL dense layers of all slots -- which nothing in the chain waits for --
//     are one launch at the end of the pass: a tile of theirs is a long contraction over the tokens, and only all of
//     them together are enough tiles to fill the chip;
//   * attention, residual + dropout + LayerNorm, the head + loss and the mini-batch gather are grouped kernels whose
//     workgroup (x, y) works on piece x of trial slot ids[y] from a per-slot descriptor;
//   * the optimiser is the per-slot fused one of optim.hip over the pack's flat parameter slab.
//
// A step of the whole pack is 14 launches per encoder layer plus 6, whatever the number of trials; the C++ driver
// below enqueues all steps of an epoch back to back and the host synchronises once per epoch.  Trials of fewer layers
// than the deepest one simply have no tiles / inactive descriptors in the deeper launches.  Dropout masks are never
// stored: forward and backward derive them from (slot seed of the epoch, site, step, element index).
//
// Every reduction runs in a fixed order (no atomics), and no kernel's arithmetic depends on which other slots are in
// the launch, so a trial's numbers are bitwise independent of its slot and of its neighbours.
#include "mt_bodies.h"

#define MT_SEQ_MAX_LAYERS 8
#define SQ_THREADS 256
#define SQ_LN_ROWS 4         // rows a wave normalises at once (forward): 16 rows per workgroup
#define SQ_LNB_ROWS 8        // rows per wave in backward: 32 rows per workgroup = one partial record
#define SQ_ZERO_ROWS 64      // rows per zero-fill workgroup of the head kernel

// ------------------------------------------------------------------------------------------------ descriptors
struct MtLnSlot {
  const float* x;        // branch output [M][d]; dropout is applied to it here
  const float* res;      // residual input [M][d]
  const float* w;
  const float* b;
  float* y;              // normalised output
  float* z;              // res + dropout(x), kept for the backward pass
  float* stats;          // [M][2] mean, rstd
  const float* dy;       // backward: gradient of y
  float* dx;             // backward: gradient of x (dropout mask applied)
  float* dres;           // backward: gradient of res
  float* part;           // backward: [ceil(M / 32)][2 d] partial sums of the weight / bias gradient
  float* gw;             // where mt_seq_ln_reduce puts the summed weight gradient
  float* gb;
  const uint32_t* seed;
  int32_t M, d, active_steps, site;
  float keep_scale, eps;
  uint32_t thresh;
  int32_t S;             // rows per sample (stochastic depth draws one keep flag per sample); 0 = no stochastic depth
  uint32_t sd_thresh;    // drop-path rate * 2^32, 0 = the branch is always kept
  float sd_scale;        // 1 / (1 - rate)
  int32_t sd_site;       // the site the per-sample flags are drawn at
  int32_t pad;
};

// Stochastic depth: the whole branch of sample b = row / S is dropped (0) or kept and scaled (sd_scale), from the
// counter hash of (slot seed, sd_site, step, b) -- forward and backward draw the same flag.
__device__ __forceinline__ float mt_ln_sample_scale(const MtLnSlot& a, bool on, uint32_t q0, uint32_t q1, int row) {
  if (!on) return 1.f;
  return mt_rng_hash(q0, q1, (uint32_t)(row / a.S)) >= a.sd_thresh ? a.sd_scale : 0.f;
}

struct MtSeqHeadSlot {
  const float* H;        // [B*S][d] output of the slot's last encoder layer
  float* dH;             // [B*S][d] out: gradient of H (zero except the last token of every sample)
  const float* w;        // [d]
  const float* b;        // [1]
  float* gw;
  float* gb;
  const float* y;        // [B] targets of the step (training) / validation targets (evaluation)
  float* loss_acc;       // += mean batch loss
  double* metrics;       // evaluation: += {sum loss, sum squared error, sum abs pct error}
  int32_t B, d, loss_type, active_steps;
};

struct MtSeqOp {          // one grouped GEMM launch
  const MtGemmGroup* groups;
  const MtTile* tiles;
  int32_t ntiles, edge;
};

struct MtAttnSlot;      // attention.hip

struct MtSeqPlan {
  int32_t n_layers, n_active, S, max_b, max_bh, max_m, max_params, wide;
  int32_t row_len, hb, use_clip, ksplits, max_blocks, pad0;
  uint32_t n_rows, pad1;
  const float* X;
  const float* Y;
  const MtGatherSlot* gather;
  const int32_t* ids;
  MtSeqOp in_proj;
  MtSeqOp qkv[MT_SEQ_MAX_LAYERS], outp[MT_SEQ_MAX_LAYERS], ff1[MT_SEQ_MAX_LAYERS], ff2[MT_SEQ_MAX_LAYERS];
  const MtAttnSlot* attn[MT_SEQ_MAX_LAYERS];
  const MtLnSlot* ln1[MT_SEQ_MAX_LAYERS];
  const MtLnSlot* ln2[MT_SEQ_MAX_LAYERS];
  MtSeqOp d_ff2[MT_SEQ_MAX_LAYERS], d_ff1[MT_SEQ_MAX_LAYERS], d_out[MT_SEQ_MAX_LAYERS], d_qkv[MT_SEQ_MAX_LAYERS];
  MtSeqOp wgrad;          // weight + bias gradients of every dense layer of every slot (mode 2, split over the tokens)
  const MtSeqHeadSlot* head;
  float* P;
  float* G;
  float* M;
  float* V;
  size_t slot_stride, gsplit_stride;
  const MtOptSlot* opt;
  float* partials;
};

extern "C" {
int mt_gemm_f32_launch(int mode, const MtGemmGroup* groups, const MtTile* tiles, int ntiles, int step, int flags,
                       int arow0, int i_override, int ksplits, size_t split_stride, int tile_edge, hipStream_t stream);
int mt_gemm_f32_bwd_launch(const MtGemmGroup* wgroups, const MtTile* wtiles, int n_wtiles, const MtGemmGroup* dgroups,
                           const MtTile* dtiles, int n_dtiles, int step, int ksplits, size_t split_stride, int tile_edge,
                           hipStream_t stream);
int mt_attn_fwd_grouped_launch(const MtAttnSlot* slots, const int32_t* ids, int n_active, int max_bh, int S, int wide,
                               int step, int b_override, int train, hipStream_t stream);
int mt_attn_bwd_grouped_launch(const MtAttnSlot* slots, const int32_t* ids, int n_active, int max_bh, int S, int wide,
                               int step, hipStream_t stream);
int mt_optim_launch(float* P, const float* G, float* M, float* V, void* Pb, size_t slot_stride, const MtOptSlot* hp,
                    const int32_t* ids, int n_active, int max_params, float* partials, int max_blocks, int use_clip,
                    int step, int gsplits, size_t gsplit_stride, hipStream_t stream);
}

// ------------------------------------------------------------------------------------------------ mini-batch gather
// Workgroup (i, y): sample i of slot ids[y]'s step -- one [S][Fpad] row of the (already padded) training table.
__global__ __launch_bounds__(SQ_THREADS) void mt_seq_gather_kernel(const MtGatherSlot* __restrict__ slots,
                                                                   const int32_t* __restrict__ ids,
                                                                   const float* __restrict__ X, const float* __restrict__ Y,
                                                                   uint32_t n_rows, int row_len, int hb, int step_imm,
                                                                   const int* __restrict__ step_base) {
  const int step = mt_step(step_imm, step_base);
  const MtGatherSlot s = slots[ids[blockIdx.y]];
  if (step >= s.active_steps) return;
  const int i = blockIdx.x;
  if (i >= s.B) return;
  const uint32_t pos = (uint32_t)step * (uint32_t)s.B + (uint32_t)i;
  const uint32_t src = mt_permute(pos, n_rows, s.key, hb);
  const f32x4* from = reinterpret_cast<const f32x4*>(X + (size_t)src * row_len);
  f32x4* to = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(s.X0) + (size_t)i * row_len);
  for (int k = threadIdx.x; k < (row_len >> 2); k += SQ_THREADS) to[k] = from[k];
  if (threadIdx.x == 0) s.y[i] = Y[src];
}

// ------------------------------------------------------------------------------------------------ LayerNorm, forward
// Rows of d <= 256 * NV floats, NV float4 per lane (columns 4 * (lane + 64 j)), SQ_LN_ROWS rows per wave with all loads
// issued before the first reduction.  Grid (ceil(max rows / 16), active slots).
template <int NV>
__global__ __launch_bounds__(SQ_THREADS) void mt_seq_ln_fwd_kernel(const MtLnSlot* __restrict__ slots,
                                                                   const int32_t* __restrict__ ids, int step_imm,
                                                                   int m_override, int train,
                                                                   const int* __restrict__ step_base) {
  const int step = mt_step(step_imm, step_base);
  const MtLnSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  const int M = m_override > 0 ? m_override : a.M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * 4 + wave) * SQ_LN_ROWS;
  if (row0 >= M) return;
  const int d = a.d;
  bool col_ok[NV];
  int c4[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) { col_ok[j] = 4 * (lane + 64 * j) < d; c4[j] = col_ok[j] ? 4 * (lane + 64 * j) : 0; }
  const bool drop = train && a.thresh != 0u && a.seed != nullptr;
  const bool sdrop = train && a.sd_thresh != 0u && a.seed != nullptr && a.S > 0;
  uint32_t s0 = 0u, s1 = 0u, q0 = 0u, q1 = 0u;
  if (drop) mt_site_seed(a.seed, a.site, step, s0, s1);
  if (sdrop) mt_site_seed(a.seed, a.sd_site, step, q0, q1);
  f32x4 v[SQ_LN_ROWS][NV];
  float sum[SQ_LN_ROWS];
#pragma unroll
  for (int i = 0; i < SQ_LN_ROWS; ++i) {
    const int rc = row0 + i < M ? row0 + i : row0;
    const float sf = mt_ln_sample_scale(a, sdrop, q0, q1, rc);
    sum[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const bool ok = col_ok[j] && (row0 + i < M);
      const size_t off = (size_t)rc * d + c4[j];
      f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + off);
      const f32x4 rv = *reinterpret_cast<const f32x4*>(a.res + off);
      if (drop) {
        const uint32_t kw = mt_keep_word4s(s0, s1, a.thresh, (uint32_t)off);
#pragma unroll
        for (int e = 0; e < 4; ++e) xv[e] *= ((kw >> (8 * e)) & 0xFFu) ? a.keep_scale : 0.f;
      }
      if (sdrop) xv = xv * sf;
      const f32x4 zv = xv + rv;
      v[i][j] = ok ? zv : f32x4{0.f, 0.f, 0.f, 0.f};
      sum[i] += (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
    }
  }
  float mean[SQ_LN_ROWS], rstd[SQ_LN_ROWS];
#pragma unroll
  for (int i = 0; i < SQ_LN_ROWS; ++i) mean[i] = wave_sum(sum[i]) / (float)d;
#pragma unroll
  for (int i = 0; i < SQ_LN_ROWS; ++i) {
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j)
      if (col_ok[j]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float c0 = v[i][j][e] - mean[i]; var += c0 * c0; }
      }
    sum[i] = var;
  }
#pragma unroll
  for (int i = 0; i < SQ_LN_ROWS; ++i) rstd[i] = rsqrtf(wave_sum(sum[i]) / (float)d + a.eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(a.w + c4[j]), bv = *reinterpret_cast<const f32x4*>(a.b + c4[j]);
#pragma unroll
    for (int i = 0; i < SQ_LN_ROWS; ++i) {
      if (col_ok[j] && row0 + i < M) {
        const size_t off = (size_t)(row0 + i) * d + c4[j];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[i][j][e] - mean[i]) * rstd[i] * wv[e] + bv[e];
        *reinterpret_cast<f32x4*>(a.z + off) = v[i][j];
        *reinterpret_cast<f32x4*>(a.y + off) = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < SQ_LN_ROWS; ++i)
    if (lane == 0 && row0 + i < M) { a.stats[2 * (size_t)(row0 + i)] = mean[i]; a.stats[2 * (size_t)(row0 + i) + 1] = rstd[i]; }
}

// ------------------------------------------------------------------------------------------------ LayerNorm, backward
// dz = rstd (g - mean(g) - xhat mean(g xhat)), g = dy w;  dres = dz, dx = dz * dropout mask * sample scale;  per-workgroup
// partial sums of dw = sum dy xhat and db = sum dy over the 32 rows of the workgroup.  Grid (ceil(max rows / 32), slots).
template <int NV>
__global__ __launch_bounds__(SQ_THREADS) void mt_seq_ln_bwd_kernel(const MtLnSlot* __restrict__ slots,
                                                                   const int32_t* __restrict__ ids, int step_imm,
                                                                   const int* __restrict__ step_base) {
  __shared__ float red[4][2 * 256 * NV];
  const int step = mt_step(step_imm, step_base);
  const MtLnSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  const int M = a.M, d = a.d;
  if ((int)blockIdx.x * 4 * SQ_LNB_ROWS >= M) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool col_ok[NV];
  int c4[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) { col_ok[j] = 4 * (lane + 64 * j) < d; c4[j] = col_ok[j] ? 4 * (lane + 64 * j) : 0; }
  const bool drop = a.thresh != 0u && a.seed != nullptr;
  const bool sdrop = a.sd_thresh != 0u && a.seed != nullptr && a.S > 0;
  uint32_t s0 = 0u, s1 = 0u, q0 = 0u, q1 = 0u;
  if (drop) mt_site_seed(a.seed, a.site, step, s0, s1);
  if (sdrop) mt_site_seed(a.seed, a.sd_site, step, q0, q1);
  f32x4 wv[NV], dw[NV], db[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    wv[j] = *reinterpret_cast<const f32x4*>(a.w + c4[j]);
    dw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[j] = dw[j];
  }
  const int row0 = (blockIdx.x * 4 + wave) * SQ_LNB_ROWS;
  const float inv_d = 1.f / (float)d;
#pragma unroll
  for (int it = 0; it < SQ_LNB_ROWS; it += SQ_LN_ROWS) {
    if (row0 + it >= M) break;                                   // wave-uniform
    f32x4 g[SQ_LN_ROWS][NV], xh[SQ_LN_ROWS][NV];
    float t1[SQ_LN_ROWS], t2[SQ_LN_ROWS], rstd[SQ_LN_ROWS];
#pragma unroll
    for (int i = 0; i < SQ_LN_ROWS; ++i) {
      const int row = row0 + it + i;
      const int rc = row < M ? row : row0 + it;                  // clamped: loads never branch
      const float mean = a.stats[2 * (size_t)rc];
      rstd[i] = a.stats[2 * (size_t)rc + 1];
      t1[i] = 0.f; t2[i] = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const bool ok = col_ok[j] && row < M;
        const size_t off = (size_t)rc * d + c4[j];
        const f32x4 dyv = *reinterpret_cast<const f32x4*>(a.dy + off);
        const f32x4 zv = *reinterpret_cast<const f32x4*>(a.z + off);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dye = ok ? dyv[e] : 0.f;
          xh[i][j][e] = ok ? (zv[e] - mean) * rstd[i] : 0.f;
          g[i][j][e] = dye * wv[j][e];
          t1[i] += g[i][j][e];
          t2[i] += g[i][j][e] * xh[i][j][e];
          dw[j][e] += dye * xh[i][j][e];
          db[j][e] += dye;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < SQ_LN_ROWS; ++i) t1[i] = wave_sum(t1[i]) * inv_d;
#pragma unroll
    for (int i = 0; i < SQ_LN_ROWS; ++i) t2[i] = wave_sum(t2[i]) * inv_d;
#pragma unroll
    for (int i = 0; i < SQ_LN_ROWS; ++i) {
      const int row = row0 + it + i;
      if (row >= M) continue;
      const float sf = mt_ln_sample_scale(a, sdrop, q0, q1, row);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (!col_ok[j]) continue;
        const size_t off = (size_t)row * d + c4[j];
        f32x4 dz, dxv;
#pragma unroll
        for (int e = 0; e < 4; ++e) dz[e] = rstd[i] * (g[i][j][e] - t1[i] - xh[i][j][e] * t2[i]);
        dxv = dz;
        if (drop) {
          const uint32_t kw = mt_keep_word4s(s0, s1, a.thresh, (uint32_t)off);
#pragma unroll
          for (int e = 0; e < 4; ++e) dxv[e] *= ((kw >> (8 * e)) & 0xFFu) ? a.keep_scale : 0.f;
        }
        if (sdrop) dxv = dxv * sf;
        *reinterpret_cast<f32x4*>(a.dres + off) = dz;
        *reinterpret_cast<f32x4*>(a.dx + off) = dxv;
      }
    }
  }
  constexpr int W = 256 * NV;
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[wave][4 * (lane + 64 * j) + e] = dw[j][e];
      red[wave][W + 4 * (lane + 64 * j) + e] = db[j][e];
    }
  __syncthreads();
  float* out = a.part + (size_t)blockIdx.x * 2 * d;
  for (int k = threadIdx.x; k < d; k += SQ_THREADS) {
    out[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    out[d + k] = (red[0][W + k] + red[1][W + k]) + (red[2][W + k] + red[3][W + k]);
  }
}

// Launchers: NV = 1 for packs whose widest slot has at most 256 columns (the arithmetic of NV = 1 and NV = 2 is the same
// for such a slot except for the order in which a lane adds its second float4, which it does not have).
extern "C" int mt_seq_ln_fwd_launch(const MtLnSlot* slots, const int32_t* ids, int n_active, int max_m, int max_d, int step,
                                    int rows, int train, hipStream_t stream) {
  if (n_active <= 0 || max_m <= 0) return 0;
  if (max_d < 4 || max_d > 512 || (max_d & 3)) return -2;
  const int m = rows > 0 ? rows : max_m;
  const dim3 grid((m + 4 * SQ_LN_ROWS - 1) / (4 * SQ_LN_ROWS), n_active), block(SQ_THREADS);
  if (max_d <= 256)
    hipLaunchKernelGGL(mt_seq_ln_fwd_kernel<1>, grid, block, 0, stream, slots, ids, step, rows, train, mt_step_base_tls);
  else
    hipLaunchKernelGGL(mt_seq_ln_fwd_kernel<2>, grid, block, 0, stream, slots, ids, step, rows, train, mt_step_base_tls);
  return (int)hipGetLastError();
}

extern "C" int mt_seq_ln_bwd_launch(const MtLnSlot* slots, const int32_t* ids, int n_active, int max_m, int max_d, int step,
                                    hipStream_t stream) {
  if (n_active <= 0 || max_m <= 0) return 0;
  if (max_d < 4 || max_d > 512 || (max_d & 3)) return -2;
  const dim3 grid((max_m + 4 * SQ_LNB_ROWS - 1) / (4 * SQ_LNB_ROWS), n_active), block(SQ_THREADS);
  if (max_d <= 256)
    hipLaunchKernelGGL(mt_seq_ln_bwd_kernel<1>, grid, block, 0, stream, slots, ids, step, mt_step_base_tls);
  else
    hipLaunchKernelGGL(mt_seq_ln_bwd_kernel<2>, grid, block, 0, stream, slots, ids, step, mt_step_base_tls);
  return (int)hipGetLastError();
}

// Sums the per-workgroup partial records of every LayerNorm of every slot in record order.  Grid (2 * layers, slots).
struct MtLnTables {
  const MtLnSlot* t[2 * MT_SEQ_MAX_LAYERS];
};

__global__ __launch_bounds__(SQ_THREADS) void mt_seq_ln_reduce_kernel(MtLnTables tabs, const int32_t* __restrict__ ids,
                                                                      int step_imm, const int* __restrict__ step_base) {
  const int step = mt_step(step_imm, step_base);
  const MtLnSlot a = tabs.t[blockIdx.x][ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  const int nb = (a.M + 4 * SQ_LNB_ROWS - 1) / (4 * SQ_LNB_ROWS), d = a.d;
  for (int k = threadIdx.x; k < 2 * d; k += SQ_THREADS) {
    float acc = 0.f;
#pragma unroll 8
    for (int c = 0; c < nb; ++c) acc += a.part[(size_t)c * 2 * d + k];
    if (k < d) a.gw[k] = acc;
    else a.gb[k - d] = acc;
  }
}

// One LayerNorm table (engines that reduce layer by layer).
extern "C" int mt_seq_ln_reduce_launch(const MtLnSlot* slots, const int32_t* ids, int n_active, int step, hipStream_t stream) {
  if (n_active <= 0) return 0;
  MtLnTables tabs = {};
  tabs.t[0] = slots;
  hipLaunchKernelGGL(mt_seq_ln_reduce_kernel, dim3(1, n_active), dim3(SQ_THREADS), 0, stream, tabs, ids, step, mt_step_base_tls);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ head + loss
// Workgroup (0, y): prediction of every sample of slot ids[y] from its last token, loss, the head's gradients and the
// gradient rows of the last tokens.  Workgroups (x >= 1, y): zero the other rows of dH, 64 rows each.
__global__ __launch_bounds__(SQ_THREADS) void mt_seq_head_kernel(const MtSeqHeadSlot* __restrict__ slots,
                                                                 const int32_t* __restrict__ ids, int S, int step_imm,
                                                                 const int* __restrict__ step_base) {
  __shared__ float dout[1024];
  __shared__ float red[4];
  const int step = mt_step(step_imm, step_base);
  const MtSeqHeadSlot s = slots[ids[blockIdx.y]];
  if (step >= s.active_steps) return;
  const int d = s.d, B = s.B, M = B * S;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x > 0) {
    const int r0 = ((int)blockIdx.x - 1) * SQ_ZERO_ROWS;
    if (r0 >= M) return;
    const int nv = d >> 2;
    for (int q = tid; q < SQ_ZERO_ROWS * nv; q += SQ_THREADS) {
      const int r = r0 + q / nv, c = q % nv;
      if (r < M && (r % S) != S - 1) *reinterpret_cast<f32x4*>(s.dH + (size_t)r * d + 4 * c) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return;
  }
  const bool col_ok = 4 * lane < d;
  const int c4 = col_ok ? 4 * lane : 0;
  const f32x4 wv = col_ok ? *reinterpret_cast<const f32x4*>(s.w + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const float bias = s.b[0];
  float lsum = 0.f;
  for (int b = wave; b < B; b += 4) {
    const size_t row = (size_t)b * S + (S - 1);
    const f32x4 h = *reinterpret_cast<const f32x4*>(s.H + row * d + c4);
    float dot = col_ok ? (h[0] * wv[0] + h[1] * wv[1]) + (h[2] * wv[2] + h[3] * wv[3]) : 0.f;
    dot = wave_sum(dot) + bias;
    float l, g;
    loss_and_grad(s.loss_type, dot, s.y[b], l, g);
    if (lane == 0) dout[b] = g / (float)B;
    lsum += l;
  }
  if (lane == 0) red[wave] = lsum;
  __syncthreads();
  if (tid == 0) {
    s.loss_acc[0] += ((red[0] + red[1]) + (red[2] + red[3])) / (float)B;
    float gb = 0.f;
    for (int b = 0; b < B; ++b) gb += dout[b];
    s.gb[0] = gb;
  }
  for (int k = tid; k < d; k += SQ_THREADS) {
    const float wk = s.w[k];
    float gw = 0.f;
    for (int b = 0; b < B; ++b) {
      const size_t at = ((size_t)b * S + (S - 1)) * d + k;
      gw += dout[b] * s.H[at];
      s.dH[at] = dout[b] * wk;
    }
    s.gw[k] = gw;
  }
}

// Evaluation: predictions of `nb` samples (a chunk of the validation set), metric sums added to the slot's record.
__global__ __launch_bounds__(SQ_THREADS) void mt_seq_head_eval_kernel(const MtSeqHeadSlot* __restrict__ slots,
                                                                      const int32_t* __restrict__ ids, int S, int nb,
                                                                      int y_off) {
  __shared__ float red[4][3];
  const MtSeqHeadSlot s = slots[ids[blockIdx.x]];
  if (s.active_steps <= 0) return;
  const int d = s.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool col_ok = 4 * lane < d;
  const int c4 = col_ok ? 4 * lane : 0;
  const f32x4 wv = col_ok ? *reinterpret_cast<const f32x4*>(s.w + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
  const float bias = s.b[0];
  float sl = 0.f, se = 0.f, sa = 0.f;
  for (int b = wave; b < nb; b += 4) {
    const size_t row = (size_t)b * S + (S - 1);
    const f32x4 h = *reinterpret_cast<const f32x4*>(s.H + row * d + c4);
    float dot = col_ok ? (h[0] * wv[0] + h[1] * wv[1]) + (h[2] * wv[2] + h[3] * wv[3]) : 0.f;
    dot = wave_sum(dot) + bias;
    const float y = s.y[y_off + b];
    float l, g;
    loss_and_grad(s.loss_type, dot, y, l, g);
    const float e = dot - y;
    sl += l; se += e * e; sa += fabsf(e) / (fabsf(y) + 1e-8f);
  }
  if (lane == 0) { red[wave][0] = sl; red[wave][1] = se; red[wave][2] = sa; }
  __syncthreads();
  if (tid < 3) s.metrics[tid] += (double)((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
}

// ------------------------------------------------------------------------------------------------ step driver
static int seq_fwd(const MtSeqOp& op, int step, int arow0, int i_override, hipStream_t stream) {
  return mt_gemm_f32_launch(0, op.groups, op.tiles, op.ntiles, step, 0, arow0, i_override, 1, 0, op.edge, stream);
}

static int seq_check(const MtSeqPlan* p) {
  if (!p || p->n_layers < 1 || p->n_layers > MT_SEQ_MAX_LAYERS) return -10;
  if (p->n_active <= 0) return 0;
  if (p->S < 1 || p->S > 128 || p->max_b < 1 || p->max_b > 1024 || p->max_m < p->S) return -11;
  if ((p->row_len & 3) || p->ksplits < 1 || p->ksplits > 127) return -12;
  if (!p->ids || !p->head || !p->in_proj.groups) return -13;
  return 1;
}

#define SQ_TRY(call) do { int rc_ = (call); if (rc_ != 0) return rc_; } while (0)

static int seq_forward(const MtSeqPlan* p, int step, int arow0, int rows, int nb, int train, hipStream_t stream) {
  SQ_TRY(seq_fwd(p->in_proj, step, arow0, rows, stream));
  for (int l = 0; l < p->n_layers; ++l) {
    SQ_TRY(seq_fwd(p->qkv[l], step, 0, rows, stream));
    SQ_TRY(mt_attn_fwd_grouped_launch(p->attn[l], p->ids, p->n_active, nb > 0 ? nb * (p->max_bh / p->max_b) : p->max_bh,
                                      p->S, p->wide, step, nb, train, stream));
    SQ_TRY(seq_fwd(p->outp[l], step, 0, rows, stream));
    SQ_TRY(mt_seq_ln_fwd_launch(p->ln1[l], p->ids, p->n_active, p->max_m, 256, step, rows, train, stream));
    SQ_TRY(seq_fwd(p->ff1[l], step, 0, rows, stream));
    SQ_TRY(seq_fwd(p->ff2[l], step, 0, rows, stream));
    SQ_TRY(mt_seq_ln_fwd_launch(p->ln2[l], p->ids, p->n_active, p->max_m, 256, step, rows, train, stream));
  }
  return (int)hipGetLastError();
}

static int seq_dgrad(const MtSeqOp& op, int step, hipStream_t stream) {
  return mt_gemm_f32_launch(1, op.groups, op.tiles, op.ntiles, step, 0, 0, 0, 1, 0, op.edge, stream);
}

// Enqueue steps [step0, step0 + nsteps) of the epoch described by the plan's per-epoch tables.  opt_step is the
// optimiser's step index inside the epoch for step0 (normally step0).
extern "C" int mt_seq_train_steps(const MtSeqPlan* p, int step0, int nsteps, hipStream_t stream) {
  const int ok = seq_check(p);
  if (ok <= 0) return ok;
  if (!p->gather || !p->X || !p->Y || !p->opt || !p->P || !p->G || !p->M || !p->V) return -13;
  const dim3 block(SQ_THREADS);
  const int zero_blocks = (p->max_m + SQ_ZERO_ROWS - 1) / SQ_ZERO_ROWS;
  MtLnTables tabs;
  for (int l = 0; l < MT_SEQ_MAX_LAYERS; ++l) {
    tabs.t[2 * l] = l < p->n_layers ? p->ln1[l] : nullptr;
    tabs.t[2 * l + 1] = l < p->n_layers ? p->ln2[l] : nullptr;
  }
  for (int step = step0; step < step0 + nsteps; ++step) {
    hipLaunchKernelGGL(mt_seq_gather_kernel, dim3(p->max_b, p->n_active), block, 0, stream, p->gather, p->ids, p->X, p->Y,
                       p->n_rows, p->row_len, p->hb, step, mt_step_base_tls);
    SQ_TRY(seq_forward(p, step, 0, 0, 0, 1, stream));
    hipLaunchKernelGGL(mt_seq_head_kernel, dim3(1 + zero_blocks, p->n_active), block, 0, stream, p->head, p->ids, p->S, step,
                       mt_step_base_tls);
    for (int l = p->n_layers - 1; l >= 0; --l) {
      SQ_TRY(mt_seq_ln_bwd_launch(p->ln2[l], p->ids, p->n_active, p->max_m, 256, step, stream));
      SQ_TRY(seq_dgrad(p->d_ff2[l], step, stream));
      SQ_TRY(seq_dgrad(p->d_ff1[l], step, stream));
      SQ_TRY(mt_seq_ln_bwd_launch(p->ln1[l], p->ids, p->n_active, p->max_m, 256, step, stream));
      SQ_TRY(seq_dgrad(p->d_out[l], step, stream));
      SQ_TRY(mt_attn_bwd_grouped_launch(p->attn[l], p->ids, p->n_active, p->max_bh, p->S, p->wide, step, stream));
      SQ_TRY(seq_dgrad(p->d_qkv[l], step, stream));
    }
    SQ_TRY(mt_gemm_f32_launch(2, p->wgrad.groups, p->wgrad.tiles, p->wgrad.ntiles, step, MT_GEMM_FLAG_DBIAS, 0, 0,
                              p->ksplits, p->gsplit_stride, p->wgrad.edge, stream));
    hipLaunchKernelGGL(mt_seq_ln_reduce_kernel, dim3(2 * p->n_layers, p->n_active), block, 0, stream, tabs, p->ids, step,
                       mt_step_base_tls);
    SQ_TRY(mt_optim_launch(p->P, p->G, p->M, p->V, nullptr, p->slot_stride, p->opt, p->ids, p->n_active, p->max_params,
                           p->partials, p->max_blocks, p->use_clip, step, p->ksplits, p->gsplit_stride, stream));
  }
  return (int)hipGetLastError();
}

// Validation pass over n_val samples in chunks of `chunk` (the plan's descriptors are the evaluation set: shared input
// table as the first GEMM's A operand, no dropout, targets = the validation targets).
extern "C" int mt_seq_eval(const MtSeqPlan* p, int n_val, int chunk, hipStream_t stream) {
  const int ok = seq_check(p);
  if (ok <= 0) return ok;
  if (chunk < 1 || chunk > p->max_b || n_val < 0) return -14;
  for (int c0 = 0; c0 < n_val; c0 += chunk) {
    const int nb = (n_val - c0 < chunk) ? n_val - c0 : chunk;
    SQ_TRY(seq_forward(p, 0, c0 * p->S, nb * p->S, nb, 0, stream));
    hipLaunchKernelGGL(mt_seq_head_eval_kernel, dim3(p->n_active), dim3(SQ_THREADS), 0, stream, p->head, p->ids, p->S, nb, c0);
  }
  return (int)hipGetLastError();
}

extern "C" int mt_seq_plan_sizeof() { return (int)sizeof(MtSeqPlan); }
extern "C" int mt_seq_ln_slot_sizeof() { return (int)sizeof(MtLnSlot); }
extern "C" int mt_seq_head_slot_sizeof() { return (int)sizeof(MtSeqHeadSlot); }
