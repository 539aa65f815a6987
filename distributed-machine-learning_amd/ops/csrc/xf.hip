// Packed training of the configurable sequence-regression Transformer (gfx950): the reference's main model family --
// input projection, sinusoidal positions + dropout, L post-norm encoder layers whose attention is softmax or
// relu-normalised ("linear") with optional local value mixing and whose feed-forward is dense or depthwise-separable,
// optional weight sharing between the layers, stochastic depth, a five-layer ReLU head on the last token, four losses,
// four optimisers, gradient clipping and a warm-up / decay schedule -- for EVERY resident trial in shared launches.
//
// Trials differ in all of that, so a step is not a fixed sequence of launches: the Python side (packed/engine_xf.py)
// compiles the pack into a PROGRAM, a flat list of MtXfOp records, each naming one grouped launch and the descriptor table
// (one record per slot, or per GEMM group) it runs over.  mt_xf_run walks the program once per optimiser step, handing the
// step index to every kernel (dropout masks, batch permutation and schedules are functions of it), with no interpreter in
// the loop and no host synchronisation.  The kernels are the grouped exact-f32 GEMM (gemm_f32.hip), grouped attention
// (attention.hip), the pack LayerNorm (seq.hip), head + loss and the optimiser of the MLP engine, and the small grouped
// kernels below: positions + dropout, depthwise convolution over time (forward, input gradient, weight gradient), row
// normalisation of explicit score matrices (attention heads too wide for the fused kernels run as grouped GEMMs around
// it), ordered sums of per-layer gradient copies (shared weights), and zero fill.
#include "mt_bodies.h"

#define XF_THREADS 256
#define XF_CONV_MAXK 7
#define XF_CONV_ROWS 64        // rows per workgroup of the convolution's weight-gradient pass

struct MtAttnSlot;   // attention.hip
struct MtLnSlot;     // seq.hip

extern "C" {
int mt_gemm_f32_launch(int mode, const MtGemmGroup* groups, const MtTile* tiles, int ntiles, int step, int flags,
                       int arow0, int i_override, int ksplits, size_t split_stride, int tile_edge, hipStream_t stream);
int mt_attn_fwd_grouped_launch(const MtAttnSlot* slots, const int32_t* ids, int n_active, int max_bh, int S, int wide,
                               int step, int b_override, int train, hipStream_t stream);
int mt_attn_bwd_grouped_launch(const MtAttnSlot* slots, const int32_t* ids, int n_active, int max_bh, int S, int wide,
                               int step, hipStream_t stream);
int mt_seq_ln_fwd_launch(const MtLnSlot* slots, const int32_t* ids, int n_active, int max_m, int max_d, int step, int rows,
                         int train, hipStream_t stream);
int mt_seq_ln_bwd_launch(const MtLnSlot* slots, const int32_t* ids, int n_active, int max_m, int max_d, int step,
                         hipStream_t stream);
int mt_seq_ln_reduce_launch(const MtLnSlot* slots, const int32_t* ids, int n_active, int step, hipStream_t stream);
int mt_head_loss_f32_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_batch, int max_width, int step,
                            float* scratch, int with_reduce, hipStream_t stream);
int mt_head_eval_f32_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_width, int row0, int nrows,
                            int max_rows, float* scratch, hipStream_t stream);
int mt_optim_launch(float* P, const float* G, float* M, float* V, void* Pb, size_t slot_stride, const MtOptSlot* hp,
                    const int32_t* ids, int n_active, int max_params, float* partials, int max_blocks, int use_clip,
                    int step, int gsplits, size_t gsplit_stride, hipStream_t stream);
}

// ------------------------------------------------------------------------------------------------ descriptors
struct MtXfGatherSlot {       // mini-batch of [S][Fpad] sample rows through the keyed permutation
  float* X0;
  float* y;
  uint32_t key[4];
  int32_t B, active_steps;
};

struct MtPeSlot {
  float* x;                   // [M][d] projected inputs; positions are added and dropout applied in place
  const float* pe;            // [S][d] sinusoid table of the slot's width
  float* dx;                  // backward: gradient of the output, masked in place
  const uint32_t* seed;
  int32_t M, S, d, active_steps;
  float keep_scale;
  uint32_t thresh;
  int32_t site, pad;
};

struct MtConvSlot {
  const float* x;             // [M][ldx] input (a column block of a wider buffer is fine)
  const float* w;             // [d][k]
  const float* b;             // [d] or null
  float* y;                   // [M][ldy] out: conv(x) + b (+ x when residual)
  const float* dy;            // backward: gradient of y, [M][lddy]
  const float* add;           // backward: added to the input gradient (a residual path's gradient), [M][lddx], or null
  float* dx;                  // backward out: mirror-conv(dy) (+ dy when residual) (+ add), [M][lddx]
  float* part;                // backward: [ceil(M / 64)][k + 1][d] partial sums of the weight / bias gradient
  float* gw;                  // [d][k]
  float* gb;                  // [d]
  int32_t M, S, d, k;
  int32_t ldx, ldy, lddy, lddx;
  int32_t residual, active_steps;
};

struct MtRowSlot {            // explicit score matrices of one slot's attention layer: [BH][S][S]
  float* sc;                  // in: raw q.k products; out: normalised probabilities (before dropout)
  float* pd;                  // out: probabilities after dropout (what multiplies V); == sc when there is no dropout
  float* stat;                // [BH][S] log-sum-exp (softmax) or the row's denominator (relu)
  float* dp;                  // backward in: gradient of the dropped probabilities; out: gradient of the raw products
  const uint32_t* seed;
  int32_t BH, S, kind, active_steps;
  int32_t H, site;
  float scale, keep_scale;
  uint32_t thresh;
  int32_t pad;
};

struct MtSumSlot {            // dst[i] = sum over p < parts of src[p * stride + i], p ascending
  const float* src;
  float* dst;
  int64_t stride;
  int32_t n, parts, active_steps, pad;
};

struct MtFillSlot {
  float* p;
  int64_t n;                  // floats, a multiple of 4
  int32_t active_steps, pad;
};

enum MtXfKind {
  XF_GATHER = 1, XF_GEMM = 2, XF_PE_FWD = 3, XF_PE_BWD = 4, XF_ATTN_FWD = 5, XF_ATTN_BWD = 6, XF_LN_FWD = 7, XF_LN_BWD = 8,
  XF_LN_REDUCE = 9, XF_CONV_FWD = 10, XF_CONV_BWD = 11, XF_CONV_WGRAD = 12, XF_ROW_FWD = 13, XF_ROW_BWD = 14, XF_SUM = 15,
  XF_FILL = 16, XF_HEAD_LOSS = 17, XF_HEAD_EVAL = 18, XF_OPTIM = 19
};

struct MtXfOp {
  int32_t kind;
  int32_t when;               // bit 0: training steps, bit 1: evaluation passes
  int32_t n;                  // GEMM: tiles; everything else: active slots in `ids`
  int32_t a, b, c, d;         // kind-specific (see mt_xf_run)
  int32_t flags;
  const void* p0;             // GEMM: groups; everything else: the slot table
  const void* p1;             // GEMM: tiles; everything else: ids
  const void* p2;             // kind-specific extra
  const void* p3;
};

struct MtXfOptim {            // what XF_OPTIM needs besides the per-slot hyper-parameter table
  float* P;
  float* G;
  float* M;
  float* V;
  float* partials;
  size_t slot_stride;
  int32_t max_params, max_blocks, use_clip, pad;
};

// ------------------------------------------------------------------------------------------------ small kernels
__device__ __forceinline__ float xf_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__global__ __launch_bounds__(XF_THREADS) void mt_xf_gather_kernel(const MtXfGatherSlot* __restrict__ slots,
                                                                  const int32_t* __restrict__ ids,
                                                                  const float* __restrict__ X, const float* __restrict__ Y,
                                                                  uint32_t n_rows, int row_len, int hb, int step) {
  const MtXfGatherSlot s = slots[ids[blockIdx.y]];
  if (step >= s.active_steps) return;
  const int i = blockIdx.x;
  if (i >= s.B) return;
  const uint32_t pos = (uint32_t)step * (uint32_t)s.B + (uint32_t)i;
  const uint32_t src = mt_permute(pos, n_rows, s.key, hb);
  const f32x4* from = reinterpret_cast<const f32x4*>(X + (size_t)src * row_len);
  f32x4* to = reinterpret_cast<f32x4*>(s.X0 + (size_t)i * row_len);
  for (int k = threadIdx.x; k < (row_len >> 2); k += XF_THREADS) to[k] = from[k];
  if (threadIdx.x == 0) s.y[i] = Y[src];
}

// x[row][c] = dropout(x[row][c] + pe[row % S][c]); BWD: dx *= the same mask.  One float4 per thread.
template <bool BWD>
__global__ __launch_bounds__(XF_THREADS) void mt_xf_pe_kernel(const MtPeSlot* __restrict__ slots, const int32_t* __restrict__ ids,
                                                              int step, int m_override, int train) {
  const MtPeSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  const int M = m_override > 0 ? m_override : a.M;
  const int d4 = a.d >> 2;
  const long i4 = (long)blockIdx.x * XF_THREADS + threadIdx.x;
  if (i4 >= (long)M * d4) return;
  const int row = (int)(i4 / d4), c = (int)(i4 % d4) * 4;
  const size_t off = (size_t)row * a.d + c;
  const bool drop = train && a.thresh != 0u && a.seed != nullptr;
  uint32_t kw = 0x01010101u;
  if (drop) {
    uint32_t s0, s1;
    mt_site_seed(a.seed, a.site, step, s0, s1);
    kw = mt_keep_word4s(s0, s1, a.thresh, (uint32_t)off);
  }
  if (BWD) {
    if (!drop) return;
    f32x4 g = *reinterpret_cast<const f32x4*>(a.dx + off);
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] *= ((kw >> (8 * e)) & 0xFFu) ? a.keep_scale : 0.f;
    *reinterpret_cast<f32x4*>(a.dx + off) = g;
  } else {
    f32x4 v = *reinterpret_cast<const f32x4*>(a.x + off) + *reinterpret_cast<const f32x4*>(a.pe + (size_t)(row % a.S) * a.d + c);
    if (drop) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= ((kw >> (8 * e)) & 0xFFu) ? a.keep_scale : 0.f;
    }
    *reinterpret_cast<f32x4*>(a.x + off) = v;
  }
}

// Depthwise convolution over the tokens of a sample, channel-last, zero padding k / 2 (same arithmetic and summation
// order as the one-trial kernel of conv1d.hip).  MIRROR: the input gradient (taps mirrored), reading dy, adding `add`.
template <bool MIRROR>
__global__ __launch_bounds__(XF_THREADS) void mt_xf_conv_kernel(const MtConvSlot* __restrict__ slots,
                                                                const int32_t* __restrict__ ids, int step, int m_override) {
  const MtConvSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  const int M = (!MIRROR && m_override > 0) ? m_override : a.M;
  const int d4 = a.d >> 2;
  const long i4 = (long)blockIdx.x * XF_THREADS + threadIdx.x;
  if (i4 >= (long)M * d4) return;
  const long row = i4 / d4;
  const int c = (int)(i4 % d4) * 4;
  const int s = (int)(row % a.S), k = a.k, half = k >> 1;
  const float* in = MIRROR ? a.dy : a.x;
  const int ldi = MIRROR ? a.lddy : a.ldx, ldo = MIRROR ? a.lddx : a.ldy;
  f32x4 acc = (!MIRROR && a.b != nullptr) ? *reinterpret_cast<const f32x4*>(a.b + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < k; ++t) {
    const int ss = s + t - half;
    if (ss < 0 || ss >= a.S) continue;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(in + (row + (t - half)) * ldi + c);
    const int tw = MIRROR ? (k - 1 - t) : t;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += a.w[(c + e) * k + tw] * xv[e];
  }
  if (a.residual) acc = acc + *reinterpret_cast<const f32x4*>(in + row * ldi + c);
  if (MIRROR && a.add != nullptr) acc = acc + *reinterpret_cast<const f32x4*>(a.add + row * ldo + c);
  *reinterpret_cast<f32x4*>((MIRROR ? a.dx : a.y) + row * ldo + c) = acc;
}

// part[chunk][k + 1][d]: taps 0 .. k-1, then the bias gradient.  Workgroup = 64 channels x 4 row lanes over 64 rows.
__global__ __launch_bounds__(XF_THREADS) void mt_xf_conv_wgrad_kernel(const MtConvSlot* __restrict__ slots,
                                                                      const int32_t* __restrict__ ids, int step) {
  const MtConvSlot a = slots[ids[blockIdx.z]];
  if (step >= a.active_steps) return;
  const long r0 = (long)blockIdx.y * XF_CONV_ROWS;
  if (r0 >= a.M || (int)blockIdx.x * 64 >= a.d) return;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int lane_r = threadIdx.x >> 6;
  const int k = a.k, half = k >> 1, d = a.d;
  float acc[XF_CONV_MAXK + 1];
#pragma unroll
  for (int t = 0; t <= XF_CONV_MAXK; ++t) acc[t] = 0.f;
  if (c < d) {
    for (long row = r0 + lane_r; row < r0 + XF_CONV_ROWS && row < a.M; row += 4) {
      const int s = (int)(row % a.S);
      const float g = a.dy[row * a.lddy + c];
      acc[XF_CONV_MAXK] += g;
#pragma unroll
      for (int t = 0; t < XF_CONV_MAXK; ++t) {
        const int ss = s + t - half;
        if (t < k && ss >= 0 && ss < a.S) acc[t] += g * a.x[(row + (t - half)) * a.ldx + c];
      }
    }
  }
  __shared__ float red[4][XF_CONV_MAXK + 1][64];
  const int cl = threadIdx.x & 63;
#pragma unroll
  for (int t = 0; t <= XF_CONV_MAXK; ++t) red[lane_r][t][cl] = acc[t];
  __syncthreads();
  if (lane_r == 0 && c < d) {
    float* out = a.part + (size_t)blockIdx.y * (k + 1) * d;
    for (int t = 0; t < k; ++t) out[(size_t)t * d + c] = (red[0][t][cl] + red[1][t][cl]) + (red[2][t][cl] + red[3][t][cl]);
    out[(size_t)k * d + c] = (red[0][XF_CONV_MAXK][cl] + red[1][XF_CONV_MAXK][cl]) +
                             (red[2][XF_CONV_MAXK][cl] + red[3][XF_CONV_MAXK][cl]);
  }
}

// gw[c][t] = sum over chunks of part[chunk][t][c], gb[c] likewise from tap index k; chunks in order.
__global__ __launch_bounds__(XF_THREADS) void mt_xf_conv_reduce_kernel(const MtConvSlot* __restrict__ slots,
                                                                       const int32_t* __restrict__ ids, int step) {
  const MtConvSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  const int d = a.d, k = a.k, chunks = (a.M + XF_CONV_ROWS - 1) / XF_CONV_ROWS;
  const int i = blockIdx.x * XF_THREADS + threadIdx.x;
  if (i >= (k + 1) * d) return;
  const int t = i / d, c = i - t * d;
  float acc = 0.f;
  for (int q = 0; q < chunks; ++q) acc += a.part[(size_t)q * (k + 1) * d + i];
  if (t < k) a.gw[c * k + t] = acc;
  else a.gb[c] = acc;
}

// One wave per row of S <= 128 scores: lane l holds columns l and l + 64.
template <bool BWD>
__global__ __launch_bounds__(XF_THREADS) void mt_xf_row_kernel(const MtRowSlot* __restrict__ slots,
                                                               const int32_t* __restrict__ ids, int step, int b_override,
                                                               int train) {
  const MtRowSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  const int BH = (!BWD && b_override > 0) ? b_override * a.H : a.BH;
  const int S = a.S;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)BH * S) return;
  const int lane = threadIdx.x & 63;
  const bool ok0 = lane < S, ok1 = lane + 64 < S;
  const size_t at = (size_t)row * S;
  const bool seeded = train && a.thresh != 0u && a.seed != nullptr;
  uint32_t s0 = 0u, s1 = 0u;
  if (seeded) mt_site_seed(a.seed, a.site, step, s0, s1);
  float k0 = 1.f, k1 = 1.f;
  if (seeded) {
    k0 = mt_rng_hash(s0, s1, (uint32_t)(at + lane)) >= a.thresh ? a.keep_scale : 0.f;
    k1 = mt_rng_hash(s0, s1, (uint32_t)(at + lane + 64)) >= a.thresh ? a.keep_scale : 0.f;
  }
  if (!BWD) {
    float v0 = ok0 ? a.sc[at + lane] * a.scale : 0.f, v1 = ok1 ? a.sc[at + lane + 64] * a.scale : 0.f;
    float p0, p1;
    if (a.kind == 0) {
      float m = fmaxf(ok0 ? v0 : -3.0e38f, ok1 ? v1 : -3.0e38f);
      m = xf_wave_max(m);
      const float e0 = ok0 ? expf(v0 - m) : 0.f, e1 = ok1 ? expf(v1 - m) : 0.f;
      const float l = wave_sum(e0 + e1);
      const float norm = 1.f / l;
      p0 = e0 * norm; p1 = e1 * norm;
      if (lane == 0) a.stat[row] = m + logf(l);
    } else {
      const float r0 = fmaxf(v0, 0.f), r1 = fmaxf(v1, 0.f);
      const float z = wave_sum((ok0 ? r0 : 0.f) + (ok1 ? r1 : 0.f)) + 1e-6f;
      const float norm = 1.f / z;
      p0 = r0 * norm; p1 = r1 * norm;
      if (lane == 0) a.stat[row] = z;
    }
    if (ok0) { a.sc[at + lane] = p0; if (a.pd != a.sc) a.pd[at + lane] = p0 * k0; }
    if (ok1) { a.sc[at + lane + 64] = p1; if (a.pd != a.sc) a.pd[at + lane + 64] = p1 * k1; }
  } else {
    const float p0 = ok0 ? a.sc[at + lane] : 0.f, p1 = ok1 ? a.sc[at + lane + 64] : 0.f;
    const float g0 = ok0 ? a.dp[at + lane] * k0 : 0.f, g1 = ok1 ? a.dp[at + lane + 64] * k1 : 0.f;
    const float delta = wave_sum(g0 * p0 + g1 * p1);
    float r0, r1;
    if (a.kind == 0) {
      r0 = p0 * (g0 - delta) * a.scale;
      r1 = p1 * (g1 - delta) * a.scale;
    } else {
      const float inv = 1.f / a.stat[row];
      r0 = p0 > 0.f ? (g0 - delta) * inv * a.scale : 0.f;
      r1 = p1 > 0.f ? (g1 - delta) * inv * a.scale : 0.f;
    }
    if (ok0) a.dp[at + lane] = r0;
    if (ok1) a.dp[at + lane + 64] = r1;
  }
}

__global__ __launch_bounds__(XF_THREADS) void mt_xf_sum_kernel(const MtSumSlot* __restrict__ slots, const int32_t* __restrict__ ids,
                                                               int step) {
  const MtSumSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  const long i = ((long)blockIdx.x * XF_THREADS + threadIdx.x) * 4;
  if (i >= a.n) return;
  f32x4 acc = *reinterpret_cast<const f32x4*>(a.src + i);
  for (int p = 1; p < a.parts; ++p) acc = acc + *reinterpret_cast<const f32x4*>(a.src + (size_t)p * a.stride + i);
  *reinterpret_cast<f32x4*>(a.dst + i) = acc;
}

__global__ __launch_bounds__(XF_THREADS) void mt_xf_fill_kernel(const MtFillSlot* __restrict__ slots,
                                                                const int32_t* __restrict__ ids, int step) {
  const MtFillSlot a = slots[ids[blockIdx.y]];
  if (step >= a.active_steps) return;
  const long i = ((long)blockIdx.x * XF_THREADS + threadIdx.x) * 4;
  if (i >= a.n) return;
  *reinterpret_cast<f32x4*>(a.p + i) = f32x4{0.f, 0.f, 0.f, 0.f};
}

// ------------------------------------------------------------------------------------------------ the interpreter
#define XF_TRY(call) do { int rc_ = (call); if (rc_ != 0) return rc_ < 0 ? rc_ - 1000 * (i + 1) : rc_; } while (0)

static inline unsigned xf_blocks(long n, int per) { return (unsigned)((n + per - 1) / per); }

// Operand conventions per kind (n = active slots unless stated):
//   GATHER      a = max batch, b = row_len, c = hb, d = n_rows;  p2 = X, p3 = Y
//   GEMM        n = tiles, a = mode, b = tile edge, c = 1 if the A operand is the shared evaluation table (row offset and
//               row count of the evaluation chunk apply), d = which row count an evaluation chunk overrides: 0 token rows,
//               1 one row per SAMPLE (head layers), 2 none (per-head score products); flags = launch flags; p2 (mode 2) =
//               floats between the partial results of two contraction splits
//   PE_*        a = max rows, b = max d
//   ATTN_*      a = max B * H, b = S, c = widest head, d = max H (evaluation: B * H of a chunk = nb * its own H, so the
//               grid is sized nb * d)
//   LN_*        a = max rows, b = max d
//   CONV_*      a = max rows, b = max d
//   ROW_*       a = max B * H, b = S, d = max H
//   SUM, FILL   a = max floats
//   HEAD_LOSS   a = max batch, b = max width; p2 = scratch
//   HEAD_EVAL   a = max rows of a chunk, b = max width; p2 = scratch
//   OPTIM       p0 = MtOptSlot table, p2 = MtXfOptim
extern "C" int mt_xf_run(const MtXfOp* ops, int n_ops, int step0, int nsteps, int train, int nb, int row0, int S,
                         hipStream_t stream) {
  if (!ops || n_ops <= 0 || nsteps <= 0) return 0;
  const dim3 block(XF_THREADS);
  const int mask = train ? 1 : 2;
  const int rows = train ? 0 : nb * S;          // evaluation: token rows of the chunk
  for (int step = step0; step < step0 + nsteps; ++step) {
    for (int i = 0; i < n_ops; ++i) {
      const MtXfOp& o = ops[i];
      if (!(o.when & mask) || o.n <= 0) continue;
      const int32_t* ids = reinterpret_cast<const int32_t*>(o.p1);
      switch (o.kind) {
        case XF_GATHER:
          hipLaunchKernelGGL(mt_xf_gather_kernel, dim3(o.a, o.n), block, 0, stream,
                             reinterpret_cast<const MtXfGatherSlot*>(o.p0), ids, reinterpret_cast<const float*>(o.p2),
                             reinterpret_cast<const float*>(o.p3), (uint32_t)o.d, o.b, o.c, step);
          break;
        case XF_GEMM: {
          const int over = (train || o.d == 2) ? 0 : (o.d == 1 ? nb : rows);
          const int arow0 = (!train && o.c) ? row0 * S : 0;
          // weight gradients: groups split over the contraction say so themselves (MtGemmGroup.site = splits); p2 carries
          // the distance in floats between two splits' partial results
          XF_TRY(mt_gemm_f32_launch(o.a, reinterpret_cast<const MtGemmGroup*>(o.p0), reinterpret_cast<const MtTile*>(o.p1), o.n,
                                    step, o.flags, arow0, over, 1, (size_t)o.p2, o.b, stream));
          break;
        }
        case XF_PE_FWD:
          hipLaunchKernelGGL(mt_xf_pe_kernel<false>, dim3(xf_blocks((long)(rows > 0 ? rows : o.a) * (o.b / 4), XF_THREADS), o.n),
                             block, 0, stream, reinterpret_cast<const MtPeSlot*>(o.p0), ids, step, rows, train);
          break;
        case XF_PE_BWD:
          hipLaunchKernelGGL(mt_xf_pe_kernel<true>, dim3(xf_blocks((long)o.a * (o.b / 4), XF_THREADS), o.n), block, 0, stream,
                             reinterpret_cast<const MtPeSlot*>(o.p0), ids, step, 0, 1);
          break;
        case XF_ATTN_FWD:
          XF_TRY(mt_attn_fwd_grouped_launch(reinterpret_cast<const MtAttnSlot*>(o.p0), ids, o.n, train ? o.a : nb * o.d, o.b,
                                            o.c, step, train ? 0 : nb, train, stream));
          break;
        case XF_ATTN_BWD:
          XF_TRY(mt_attn_bwd_grouped_launch(reinterpret_cast<const MtAttnSlot*>(o.p0), ids, o.n, o.a, o.b, o.c, step, stream));
          break;
        case XF_LN_FWD:
          XF_TRY(mt_seq_ln_fwd_launch(reinterpret_cast<const MtLnSlot*>(o.p0), ids, o.n, o.a, o.b, step, rows, train, stream));
          break;
        case XF_LN_BWD:
          XF_TRY(mt_seq_ln_bwd_launch(reinterpret_cast<const MtLnSlot*>(o.p0), ids, o.n, o.a, o.b, step, stream));
          break;
        case XF_LN_REDUCE:
          XF_TRY(mt_seq_ln_reduce_launch(reinterpret_cast<const MtLnSlot*>(o.p0), ids, o.n, step, stream));
          break;
        case XF_CONV_FWD:
          hipLaunchKernelGGL(mt_xf_conv_kernel<false>, dim3(xf_blocks((long)(rows > 0 ? rows : o.a) * (o.b / 4), XF_THREADS), o.n),
                             block, 0, stream, reinterpret_cast<const MtConvSlot*>(o.p0), ids, step, rows);
          break;
        case XF_CONV_BWD:
          hipLaunchKernelGGL(mt_xf_conv_kernel<true>, dim3(xf_blocks((long)o.a * (o.b / 4), XF_THREADS), o.n), block, 0, stream,
                             reinterpret_cast<const MtConvSlot*>(o.p0), ids, step, 0);
          break;
        case XF_CONV_WGRAD:
          hipLaunchKernelGGL(mt_xf_conv_wgrad_kernel, dim3((o.b + 63) / 64, xf_blocks(o.a, XF_CONV_ROWS), o.n), block, 0, stream,
                             reinterpret_cast<const MtConvSlot*>(o.p0), ids, step);
          hipLaunchKernelGGL(mt_xf_conv_reduce_kernel, dim3(xf_blocks((long)(XF_CONV_MAXK + 1) * o.b, XF_THREADS), o.n), block, 0,
                             stream, reinterpret_cast<const MtConvSlot*>(o.p0), ids, step);
          break;
        case XF_ROW_FWD:
          hipLaunchKernelGGL(mt_xf_row_kernel<false>, dim3(xf_blocks((long)(train ? o.a : nb * o.d) * o.b, 4), o.n), block, 0, stream,
                             reinterpret_cast<const MtRowSlot*>(o.p0), ids, step, train ? 0 : nb, train);
          break;
        case XF_ROW_BWD:
          hipLaunchKernelGGL(mt_xf_row_kernel<true>, dim3(xf_blocks((long)o.a * o.b, 4), o.n), block, 0, stream,
                             reinterpret_cast<const MtRowSlot*>(o.p0), ids, step, 0, 1);
          break;
        case XF_SUM:
          hipLaunchKernelGGL(mt_xf_sum_kernel, dim3(xf_blocks(o.a, 4 * XF_THREADS), o.n), block, 0, stream,
                             reinterpret_cast<const MtSumSlot*>(o.p0), ids, step);
          break;
        case XF_FILL:
          hipLaunchKernelGGL(mt_xf_fill_kernel, dim3(xf_blocks(o.a, 4 * XF_THREADS), o.n), block, 0, stream,
                             reinterpret_cast<const MtFillSlot*>(o.p0), ids, step);
          break;
        case XF_HEAD_LOSS:
          XF_TRY(mt_head_loss_f32_launch(reinterpret_cast<const MtHeadSlot*>(o.p0), ids, o.n, o.a, o.b, step,
                                         reinterpret_cast<float*>(const_cast<void*>(o.p2)), 1, stream));
          break;
        case XF_HEAD_EVAL:
          XF_TRY(mt_head_eval_f32_launch(reinterpret_cast<const MtHeadSlot*>(o.p0), ids, o.n, o.b, row0, nb, o.a,
                                         reinterpret_cast<float*>(const_cast<void*>(o.p2)), stream));
          break;
        case XF_OPTIM: {
          const MtXfOptim* q = reinterpret_cast<const MtXfOptim*>(o.p2);
          XF_TRY(mt_optim_launch(q->P, q->G, q->M, q->V, nullptr, q->slot_stride, reinterpret_cast<const MtOptSlot*>(o.p0), ids,
                                 o.n, q->max_params, q->partials, q->max_blocks, q->use_clip, step, 1, 0, stream));
          break;
        }
        default:
          return -900 - i;
      }
    }
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

extern "C" int mt_xf_sizes(int* out) {
  out[0] = (int)sizeof(MtXfOp);
  out[1] = (int)sizeof(MtPeSlot);
  out[2] = (int)sizeof(MtConvSlot);
  out[3] = (int)sizeof(MtRowSlot);
  out[4] = (int)sizeof(MtSumSlot);
  out[5] = (int)sizeof(MtFillSlot);
  out[6] = (int)sizeof(MtXfGatherSlot);
  out[7] = (int)sizeof(MtXfOptim);
  return 8;
}
