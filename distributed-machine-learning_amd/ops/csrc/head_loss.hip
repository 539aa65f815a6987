// Fused regression head + loss + loss gradient (training) and head + metric reduction (evaluation).
//
// Training kernel, one workgroup per trial slot.  For every batch row it computes
//   out = h . w + b,  loss(out, y),  dout = dloss/dout / B
// and from there, in the same pass over h,
//   dH[i][k] = dout * w[k] * (h[i][k] > 0)      (gradient entering the last hidden layer, ReLU folded in)
//   gw[k]   += dout * h[i][k],   gb += dout,    loss_acc += mean loss.
// The N = 1 head GEMV, the loss forward/backward, both head gradients and the per-trial loss reduction
// are therefore one launch and one read of h; nothing is synchronised with the host.  Reductions run
// in a fixed order (lane shuffle -> wave -> LDS), so results do not depend on which slot a trial is in.
//
// Evaluation kernel: same head forward, then adds {sum loss, sum squared error, sum |err|/(|y|+eps)}
// for the chunk into the slot's f64 metric accumulators -- the table the scheduler all-gathers.
#include "mt_head_body.h"

template <int NCHUNK, bool F32>
__global__ __launch_bounds__(HL_THREADS) void mt_head_loss_kernel(const MtHeadSlot* __restrict__ slots,
                                                                  const int32_t* __restrict__ ids, int step_imm,
                                                                  float* __restrict__ scratch, int max_chunks,
                                                                  int rec, const int* __restrict__ step_base) {
  __shared__ float red[HL_WAVES][NCHUNK * 512];
  __shared__ float red_s[HL_WAVES][2];
  mt_head_loss_body<NCHUNK, F32>(slots, ids, mt_step(step_imm, step_base), scratch, max_chunks, rec, blockIdx.x, blockIdx.y,
                                 red, red_s);
}

__global__ __launch_bounds__(HL_THREADS) void mt_head_reduce_kernel(const MtHeadSlot* __restrict__ slots,
                                                                    const int32_t* __restrict__ ids, int step_imm,
                                                                    const float* __restrict__ scratch,
                                                                    int max_chunks, int rec,
                                                                    const int* __restrict__ step_base) {
  mt_head_reduce_body(slots, ids, mt_step(step_imm, step_base), scratch, max_chunks, rec, blockIdx.x);
}

// Evaluation: grid (row_chunks, active slots); each workgroup writes {sum loss, sum sq err, sum abs pct err}
// of its HL_EVAL_ROWS rows to scratch, mt_head_eval_reduce_kernel adds them (chunk order) into the f64 table.
#define HL_EVAL_ROWS 256

template <int NCHUNK, bool F32>
__global__ __launch_bounds__(HL_THREADS) void mt_head_eval_kernel(const MtHeadSlot* __restrict__ slots,
                                                                  const int32_t* __restrict__ ids, int row0, int nrows,
                                                                  float* __restrict__ scratch, int max_chunks) {
  const int slot = ids[blockIdx.y];
  MtHeadSlot s = slots[slot];
  if (s.active_steps <= 0) return;
  if (nrows > 0) {   // chunked evaluation: targets [row0, row0 + nrows) of the shared validation vector
    s.B = nrows;
    s.y += row0;
  }
  const int row_lo = blockIdx.x * HL_EVAL_ROWS;
  if (row_lo >= s.B) return;
  const int row_hi = (row_lo + HL_EVAL_ROWS < s.B) ? row_lo + HL_EVAL_ROWS : s.B;
  __shared__ float red_s[HL_WAVES][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const void* H = s.H;
  float w[NCHUNK][8];
#pragma unroll
  for (int c = 0; c < NCHUNK; ++c)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = c * 512 + lane * 8 + e;
      w[c][e] = (k < s.W) ? s.w[k] : 0.f;
    }
  const float bias = s.b[0];
  float sl = 0.f, sse = 0.f, sape = 0.f;
  for (int i = row_lo + wave; i < row_hi; i += HL_WAVES) {
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) {
      const int k0 = c * 512 + lane * 8;
      HRow8<F32> hv;
      hv.load(H, (size_t)i * s.ldh + k0, k0 < s.W);
#pragma unroll
      for (int e = 0; e < 8; ++e) dot += hv.get(e) * w[c][e];
    }
    const float out = wave_sum(dot) + bias;
    const float y = s.y[i];
    float l, g;
    loss_and_grad(s.loss_type, out, y, l, g);
    const float d = out - y;
    sl += l;
    sse += d * d;
    sape += fabsf(d) / (fabsf(y) + 1e-8f);
  }
  if (lane == 0) { red_s[wave][0] = sl; red_s[wave][1] = sse; red_s[wave][2] = sape; }
  __syncthreads();
  if (tid < 3)
    scratch[((size_t)slot * max_chunks + blockIdx.x) * 4 + tid] =
        (red_s[0][tid] + red_s[1][tid]) + (red_s[2][tid] + red_s[3][tid]);
}

__global__ __launch_bounds__(64) void mt_head_eval_reduce_kernel(const MtHeadSlot* __restrict__ slots,
                                                                 const int32_t* __restrict__ ids, int nrows,
                                                                 const float* __restrict__ scratch, int max_chunks) {
  const int slot = ids[blockIdx.x];
  const MtHeadSlot s = slots[slot];
  if (s.active_steps <= 0) return;
  const int rows = nrows > 0 ? nrows : s.B;
  const int nchunk = (rows + HL_EVAL_ROWS - 1) / HL_EVAL_ROWS;
  if (threadIdx.x < 3) {
    double acc = 0.0;
    for (int c = 0; c < nchunk; ++c) acc += (double)scratch[((size_t)slot * max_chunks + c) * 4 + threadIdx.x];
    s.metrics[threadIdx.x] += acc;
  }
}


// ---- wide layers (W > 2048) ---------------------------------------------------------------------------------------------
// The register-resident forms above keep a lane's slice of w and gw for the whole row chunk, which stops fitting past
// 2048 columns.  For wider heads (data-parallel trials: 8192 columns) the same work is split in two passes over the
// chunk's HL_ROWS rows: (1) one wave per row walks the row in 512-column steps and produces out / loss / dout, which
// go to LDS; (2) every thread owns 8 columns at a time and walks the rows: gw partials need no reduction across waves
// at all (a column belongs to one thread) and dH is written on the way.  The second pass re-reads the rows from L2.
template <bool F32>
__device__ __forceinline__ float head_row_dot(const void* H, size_t row_off, const float* __restrict__ w, int W, int lane) {
  float dot = 0.f;
  for (int k0 = lane * 8; k0 < W; k0 += 512) {
    HRow8<F32> hv;
    hv.load(H, row_off + k0, true);
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(w + k0), w1 = *reinterpret_cast<const f32x4*>(w + k0 + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) dot += hv.get(e) * w0[e];
#pragma unroll
    for (int e = 0; e < 4; ++e) dot += hv.get(4 + e) * w1[e];
  }
  return wave_sum(dot);
}

template <bool F32>      // W % 8 == 0, head weights 16-byte aligned
__global__ __launch_bounds__(HL_THREADS) void mt_head_loss_wide_kernel(const MtHeadSlot* __restrict__ slots,
                                                                       const int32_t* __restrict__ ids, int step_imm,
                                                                       float* __restrict__ scratch, int max_chunks, int rec,
                                                                       const int* __restrict__ step_base) {
  const int step = mt_step(step_imm, step_base);
  const int slot = ids[blockIdx.y];
  const MtHeadSlot s = slots[slot];
  if (step >= s.active_steps) return;
  const int row_lo = blockIdx.x * HL_ROWS;
  if (row_lo >= s.B) return;
  const int row_hi = (row_lo + HL_ROWS < s.B) ? row_lo + HL_ROWS : s.B;
  __shared__ float dout_s[HL_ROWS];
  __shared__ float loss_s[HL_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float bias = s.b[0];
  const float inv_b = 1.f / (float)s.B;
  for (int i = row_lo + wave; i < row_hi; i += HL_WAVES) {
    const float out = head_row_dot<F32>(s.H, (size_t)i * s.ldh, s.w, s.W, lane) + bias;
    float l, g;
    loss_and_grad(s.loss_type, out, s.y[i], l, g);
    if (lane == 0) { dout_s[i - row_lo] = g * inv_b; loss_s[i - row_lo] = l; }
  }
  __syncthreads();
  float* out_rec = scratch + ((size_t)slot * max_chunks + blockIdx.x) * rec;
  const int nrows = row_hi - row_lo;
  for (int k0 = tid * 8; k0 < s.W; k0 += HL_THREADS * 8) {
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(s.w + k0), w1 = *reinterpret_cast<const f32x4*>(s.w + k0 + 4);
    float wv[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
    float gw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < nrows; ++r) {
      const size_t off = (size_t)(row_lo + r) * s.ldh + k0;
      HRow8<F32> hv;
      hv.load(s.H, off, true);
      const float dout = dout_s[r];
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float h = hv.get(e);
        gw[e] += dout * h;
        o[e] = h > 0.f ? dout * wv[e] : 0.f;
      }
      HRow8<F32>::store(s.dH, off, o);
    }
    *reinterpret_cast<f32x4*>(out_rec + k0) = f32x4{gw[0], gw[1], gw[2], gw[3]};
    *reinterpret_cast<f32x4*>(out_rec + k0 + 4) = f32x4{gw[4], gw[5], gw[6], gw[7]};
  }
  if (tid == 0) {
    float gb = 0.f, ls = 0.f;
    for (int r = 0; r < nrows; ++r) { gb += dout_s[r]; ls += loss_s[r]; }      // row order: fixed
    out_rec[rec - 2] = gb;
    out_rec[rec - 1] = ls * inv_b;
  }
}

template <bool F32>
__global__ __launch_bounds__(HL_THREADS) void mt_head_eval_wide_kernel(const MtHeadSlot* __restrict__ slots,
                                                                       const int32_t* __restrict__ ids, int row0, int nrows,
                                                                       float* __restrict__ scratch, int max_chunks) {
  const int slot = ids[blockIdx.y];
  MtHeadSlot s = slots[slot];
  if (s.active_steps <= 0) return;
  if (nrows > 0) {
    s.B = nrows;
    s.y += row0;
  }
  const int row_lo = blockIdx.x * HL_EVAL_ROWS;
  if (row_lo >= s.B) return;
  const int row_hi = (row_lo + HL_EVAL_ROWS < s.B) ? row_lo + HL_EVAL_ROWS : s.B;
  __shared__ float red_s[HL_WAVES][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float bias = s.b[0];
  float sl = 0.f, sse = 0.f, sape = 0.f;
  for (int i = row_lo + wave; i < row_hi; i += HL_WAVES) {
    const float out = head_row_dot<F32>(s.H, (size_t)i * s.ldh, s.w, s.W, lane) + bias;
    const float y = s.y[i];
    float l, g;
    loss_and_grad(s.loss_type, out, y, l, g);
    const float d = out - y;
    sl += l;
    sse += d * d;
    sape += fabsf(d) / (fabsf(y) + 1e-8f);
  }
  if (lane == 0) { red_s[wave][0] = sl; red_s[wave][1] = sse; red_s[wave][2] = sape; }
  __syncthreads();
  if (tid < 3)
    scratch[((size_t)slot * max_chunks + blockIdx.x) * 4 + tid] =
        (red_s[0][tid] + red_s[1][tid]) + (red_s[2][tid] + red_s[3][tid]);
}

// Scratch floats needed for n_slots physical slots: training records and evaluation records share it.
extern "C" long mt_head_scratch_floats(int n_slots, int max_rows, int max_width) {
  const long rec = ((max_width + 7) / 8) * 8 + 8;
  const long train = (long)((max_rows + HL_ROWS - 1) / HL_ROWS) * rec;
  const long eval = (long)((max_rows + HL_EVAL_ROWS - 1) / HL_EVAL_ROWS) * 4;
  return (long)n_slots * (train > eval ? train : eval);
}

// ids: the n_active physical slot numbers to process; scratch: mt_head_scratch_floats(n_slots_total, ...) floats.
// with_reduce = 0 leaves the partial records in scratch for a later launch to add (mt_gemm_bwd_head_launch).
// f32: the activation buffers H / dH hold f32 instead of bf16.
template <bool F32>
static int head_loss_launch_t(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_batch, int max_width,
                              int step, float* scratch, int with_reduce, hipStream_t stream) {
  if (n_active <= 0) return 0;
  const int rec = ((max_width + 7) / 8) * 8 + 8;
  const int max_chunks = (max_batch + HL_ROWS - 1) / HL_ROWS;
  dim3 grid(max_chunks, n_active), block(HL_THREADS);
  if (max_width <= 512)
    hipLaunchKernelGGL((mt_head_loss_kernel<1, F32>), grid, block, 0, stream, slots, ids, step, scratch, max_chunks, rec,
                       mt_step_base_tls);
  else if (max_width <= 1024)
    hipLaunchKernelGGL((mt_head_loss_kernel<2, F32>), grid, block, 0, stream, slots, ids, step, scratch, max_chunks, rec,
                       mt_step_base_tls);
  else if (max_width <= 2048)
    hipLaunchKernelGGL((mt_head_loss_kernel<4, F32>), grid, block, 0, stream, slots, ids, step, scratch, max_chunks, rec,
                       mt_step_base_tls);
  else if (max_width % 8 == 0)       // every slot of such a launch must have W % 8 == 0 (data-parallel trials do)
    hipLaunchKernelGGL((mt_head_loss_wide_kernel<F32>), grid, block, 0, stream, slots, ids, step, scratch, max_chunks, rec,
                       mt_step_base_tls);
  else return -2;
  if (with_reduce)
    hipLaunchKernelGGL(mt_head_reduce_kernel, dim3(n_active), block, 0, stream, slots, ids, step, scratch, max_chunks, rec,
                       mt_step_base_tls);
  return (int)hipGetLastError();
}

extern "C" int mt_head_loss_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_batch,
                                   int max_width, int step, float* scratch, hipStream_t stream) {
  return head_loss_launch_t<false>(slots, ids, n_active, max_batch, max_width, step, scratch, 1, stream);
}

extern "C" int mt_head_loss_partials_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_batch,
                                            int max_width, int step, float* scratch, hipStream_t stream) {
  return head_loss_launch_t<false>(slots, ids, n_active, max_batch, max_width, step, scratch, 0, stream);
}

// f32 activations; with_reduce as above.
extern "C" int mt_head_loss_f32_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_batch,
                                       int max_width, int step, float* scratch, int with_reduce, hipStream_t stream) {
  return head_loss_launch_t<true>(slots, ids, n_active, max_batch, max_width, step, scratch, with_reduce, stream);
}

template <bool F32>
static int head_eval_launch_t(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_width, int row0,
                              int nrows, int max_rows, float* scratch, hipStream_t stream) {
  if (n_active <= 0) return 0;
  const int rows = nrows > 0 ? nrows : max_rows;
  const int max_chunks = (max_rows + HL_EVAL_ROWS - 1) / HL_EVAL_ROWS;
  dim3 grid((rows + HL_EVAL_ROWS - 1) / HL_EVAL_ROWS, n_active), block(HL_THREADS);
  if (max_width <= 512)
    hipLaunchKernelGGL((mt_head_eval_kernel<1, F32>), grid, block, 0, stream, slots, ids, row0, nrows, scratch, max_chunks);
  else if (max_width <= 1024)
    hipLaunchKernelGGL((mt_head_eval_kernel<2, F32>), grid, block, 0, stream, slots, ids, row0, nrows, scratch, max_chunks);
  else if (max_width <= 2048)
    hipLaunchKernelGGL((mt_head_eval_kernel<4, F32>), grid, block, 0, stream, slots, ids, row0, nrows, scratch, max_chunks);
  else if (max_width % 8 == 0)
    hipLaunchKernelGGL((mt_head_eval_wide_kernel<F32>), grid, block, 0, stream, slots, ids, row0, nrows, scratch, max_chunks);
  else return -2;
  hipLaunchKernelGGL(mt_head_eval_reduce_kernel, dim3(n_active), dim3(64), 0, stream, slots, ids, nrows, scratch,
                     max_chunks);
  return (int)hipGetLastError();
}

extern "C" int mt_head_eval_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_width,
                                   int row0, int nrows, int max_rows, float* scratch, hipStream_t stream) {
  return head_eval_launch_t<false>(slots, ids, n_active, max_width, row0, nrows, max_rows, scratch, stream);
}

extern "C" int mt_head_eval_f32_launch(const MtHeadSlot* slots, const int32_t* ids, int n_active, int max_width,
                                       int row0, int nrows, int max_rows, float* scratch, hipStream_t stream) {
  return head_eval_launch_t<true>(slots, ids, n_active, max_width, row0, nrows, max_rows, scratch, stream);
}
