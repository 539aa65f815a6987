// Feature tokenizer of the tabular Transformer, and column sums of a row-major matrix.
//
// Forward: out[b, t, :] = x[b, t] * W[t, :] + bias[t, :] for the F feature tokens and out[b, F, :] = cls, written once as
// the f32 residual stream and (optionally) once more as the bf16 operand of the first projection -- as separate
// element-wise passes this is a multiply, an add, a concatenation and a cast over [B, F + 1, d].
// Backward: dW[t, :] = sum_b x[b, t] * g[b, t, :], dbias[t, :] = sum_b g[b, t, :], dcls = sum_b g[b, F, :], where g is the
// sum of the gradients of the two outputs.  The batch is cut into `nsplit` slices, every (slice, column) pair is one
// thread's serial sum, and the slices are added in index order afterwards (mt_slab_sum_launch): a fixed order of
// additions, no atomics.
#include "mt_common.h"

#define TK_THREADS 256

__global__ __launch_bounds__(TK_THREADS) void mt_tokens_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ cls, float* __restrict__ out,
                                                                   uint16_t* __restrict__ out16, long B, int F, int d) {
  const int d4 = d >> 2;
  const long total = B * (long)(F + 1) * d4;
  for (long e = (long)blockIdx.x * TK_THREADS + threadIdx.x; e < total; e += (long)gridDim.x * TK_THREADS) {
    const long row = e / d4;
    const int c = (int)(e - row * d4) * 4;
    const long b = row / (F + 1);
    const int t = (int)(row - b * (F + 1));
    f32x4 v;
    if (t < F) {
#pragma clang fp contract(off)
      const float xv = x[b * F + t];
      const f32x4 w = *reinterpret_cast<const f32x4*>(W + (size_t)t * d + c);
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bias + (size_t)t * d + c);
      // product and sum rounded separately (no contraction in this block), as the two tensor operations this replaces
      v = f32x4{xv * w[0] + bb[0], xv * w[1] + bb[1], xv * w[2] + bb[2], xv * w[3] + bb[3]};
    } else {
      v = *reinterpret_cast<const f32x4*>(cls + c);
    }
    *reinterpret_cast<f32x4*>(out + (size_t)row * d + c) = v;
    if (out16 != nullptr) {
      uint32_t lo = pack_bf16x2(v[0], v[1]), hi = pack_bf16x2(v[2], v[3]);
      *reinterpret_cast<uint2*>(out16 + (size_t)row * d + c) = make_uint2(lo, hi);
    }
  }
}

// part[s][0][col] = sum over the slice's batch rows of g[b][col]; part[s][1][col] = the same sum weighted by x[b, t]
// (columns of the summary token: 0).  col runs over (F + 1) * d.
__global__ __launch_bounds__(TK_THREADS) void mt_tokens_bwd_kernel(const float* __restrict__ g32,
                                                                   const uint16_t* __restrict__ g16,
                                                                   const float* __restrict__ x, float* __restrict__ part,
                                                                   long B, int F, int d, int nsplit) {
  const long cols = (long)(F + 1) * d;
  const long col = (long)blockIdx.x * TK_THREADS + threadIdx.x;
  const int s = blockIdx.y;
  if (col >= cols) return;
  const int t = (int)(col / d);
  const long per = (B + nsplit - 1) / nsplit;
  const long b0 = s * per, b1 = (b0 + per < B) ? b0 + per : B;
  float sb = 0.f, sw = 0.f;
  for (long b = b0; b < b1; ++b) {
    float g = 0.f;
    if (g32 != nullptr) g = g32[(size_t)b * cols + col];
    if (g16 != nullptr) g += bf16_bits_to_f32(g16[(size_t)b * cols + col]);
    sb += g;
    if (t < F) sw += x[b * F + t] * g;
  }
  part[((size_t)s * 2 + 0) * cols + col] = sb;
  part[((size_t)s * 2 + 1) * cols + col] = sw;
}

extern "C" int mt_tokens_fwd_launch(const float* x, const float* W, const float* bias, const float* cls, float* out,
                                    void* out16, long B, int F, int d, hipStream_t stream) {
  if (B <= 0) return 0;
  if (F < 1 || d < 4 || (d & 3)) return -2;
  const long total = B * (long)(F + 1) * (d >> 2);
  long blocks = (total + TK_THREADS - 1) / TK_THREADS;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(mt_tokens_fwd_kernel, dim3((unsigned)blocks), dim3(TK_THREADS), 0, stream, x, W, bias, cls, out,
                     (uint16_t*)out16, B, F, d);
  return (int)hipGetLastError();
}

// part: [nsplit][2][(F + 1) * d] floats.
extern "C" int mt_tokens_bwd_launch(const float* g32, const void* g16, const float* x, float* part, long B, int F, int d,
                                    int nsplit, hipStream_t stream) {
  if (B <= 0) return 0;
  if (F < 1 || d < 1 || nsplit < 1 || nsplit > 1024 || (g32 == nullptr && g16 == nullptr)) return -2;
  const long cols = (long)(F + 1) * d;
  hipLaunchKernelGGL(mt_tokens_bwd_kernel, dim3((unsigned)((cols + TK_THREADS - 1) / TK_THREADS), (unsigned)nsplit),
                     dim3(TK_THREADS), 0, stream, g32, (const uint16_t*)g16, x, part, B, F, d, nsplit);
  return (int)hipGetLastError();
}

// out[c] = sum over rows of src[r][c] (row-major, rows x cols).  A workgroup owns 16 columns; its 16 row phases each add
// their rows (phase, phase + 16, ...) in index order, four loads in flight, and the phases are then added in order 0..15:
// a fixed tree, so the result does not depend on the launch.
__global__ __launch_bounds__(TK_THREADS) void mt_colsum_kernel(const float* __restrict__ src, float* __restrict__ out,
                                                               long rows, int cols) {
  __shared__ float red[16][17];
  const int tx = threadIdx.x & 15, ph = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + tx;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    long r = ph;
    for (; r + 48 < rows; r += 64) {
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] += src[(size_t)(r + 16 * u) * cols + c];
    }
    for (; r < rows; r += 16) a[0] += src[(size_t)r * cols + c];
  }
  red[ph][tx] = (a[0] + a[1]) + (a[2] + a[3]);
  __syncthreads();
  if (ph == 0 && c < cols) {
    float t = red[0][tx];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += red[q][tx];
    out[c] = t;
  }
}

extern "C" int mt_colsum_launch(const float* src, float* out, long rows, int cols, hipStream_t stream) {
  if (cols <= 0) return 0;
  if (rows < 0) return -2;
  hipLaunchKernelGGL(mt_colsum_kernel, dim3((unsigned)((cols + 15) / 16)), dim3(TK_THREADS), 0, stream, src, out, rows, cols);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Regression head on the summary token: pred[b] = w . relu(LayerNorm(x[b]) * gamma + beta) + b0, rows of at most 512
// columns.  One wave per row; as separate tensor operations this is a LayerNorm, a clamp, a multiply, a row sum and an
// add going forward and a dozen small launches going back.  Backward writes dx and, per workgroup, partial sums of the
// four parameter gradients ([3][d] + 1 floats, rows added in index order); mt_colsum_launch adds the workgroups.
#define CH_MAXV 8          // columns per lane: d <= 512

__global__ __launch_bounds__(TK_THREADS) void mt_cls_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta,
                                                                     const float* __restrict__ w, const float* __restrict__ b0,
                                                                     float* __restrict__ pred, float* __restrict__ stats,
                                                                     int B, int d, float eps) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  float v[CH_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CH_MAXV; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < d ? x[(size_t)row * d + c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CH_MAXV; ++i) {
    const int c = lane + 64 * i;
    const float t = c < d ? v[i] - mean : 0.f;
    q += t * t;
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)d + eps);
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < CH_MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < d) {
      const float a = (v[i] - mean) * rstd * gamma[c] + beta[c];
      acc += fmaxf(a, 0.f) * w[c];
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    pred[row] = acc + b0[0];
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
  }
}

// part: [gridDim.x][3 * d + 1]: d gamma | d beta | d w | d b0
__global__ __launch_bounds__(TK_THREADS) void mt_cls_head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta,
                                                                     const float* __restrict__ w,
                                                                     const float* __restrict__ stats,
                                                                     const float* __restrict__ dpred, float* __restrict__ dx,
                                                                     float* __restrict__ part, int B, int d, int rows_per_wg) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = blockIdx.x * rows_per_wg;
  float gg[CH_MAXV], gb[CH_MAXV], gw[CH_MAXV], ga[CH_MAXV], wv[CH_MAXV], bv[CH_MAXV];
#pragma unroll
  for (int i = 0; i < CH_MAXV; ++i) {
    const int c = lane + 64 * i;
    gg[i] = gb[i] = gw[i] = 0.f;
    ga[i] = c < d ? gamma[c] : 0.f;
    wv[i] = c < d ? w[c] : 0.f;
    bv[i] = c < d ? beta[c] : 0.f;
  }
  float g0 = 0.f;
  for (int row = r0 + wave; row < r0 + rows_per_wg && row < B; row += 4) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1], dp = dpred[row];
    float z[CH_MAXV], dz[CH_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CH_MAXV; ++i) {
      const int c = lane + 64 * i;
      z[i] = c < d ? (x[(size_t)row * d + c] - mean) * rstd : 0.f;
      const float a = z[i] * ga[i] + bv[i];
      const float h = fmaxf(a, 0.f);
      const float dh = (c < d && a > 0.f) ? dp * wv[i] : 0.f;
      gw[i] += dp * h;
      gb[i] += dh;
      gg[i] += dh * z[i];
      dz[i] = dh * ga[i];
      s1 += dz[i];
      s2 += dz[i] * z[i];
    }
    s1 = wave_sum(s1) / (float)d;
    s2 = wave_sum(s2) / (float)d;
#pragma unroll
    for (int i = 0; i < CH_MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < d) dx[(size_t)row * d + c] = rstd * (dz[i] - s1 - z[i] * s2);
    }
    g0 += dp;
  }
  // the four waves' sums are added in wave order, one quantity at a time (the LDS image holds one quantity of four waves)
  float* out = part + (size_t)blockIdx.x * (3 * d + 1);
  __shared__ float img4[4][64 * CH_MAXV];
  for (int qn = 0; qn < 3; ++qn) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < CH_MAXV; ++i) img4[wave][lane + 64 * i] = qn == 0 ? gg[i] : (qn == 1 ? gb[i] : gw[i]);
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += TK_THREADS)
      out[qn * d + c] = ((img4[0][c] + img4[1][c]) + img4[2][c]) + img4[3][c];
  }
  __syncthreads();
  if (lane == 0) img4[wave][0] = g0;
  __syncthreads();
  if (threadIdx.x == 0) out[3 * d] = ((img4[0][0] + img4[1][0]) + img4[2][0]) + img4[3][0];
}

extern "C" int mt_cls_head_fwd_launch(const float* x, const float* gamma, const float* beta, const float* w, const float* b0,
                                      float* pred, float* stats, int B, int d, float eps, hipStream_t stream) {
  if (B <= 0) return 0;
  if (d < 1 || d > 64 * CH_MAXV) return -2;
  hipLaunchKernelGGL(mt_cls_head_fwd_kernel, dim3((unsigned)((B + 3) / 4)), dim3(TK_THREADS), 0, stream, x, gamma, beta, w, b0,
                     pred, stats, B, d, eps);
  return (int)hipGetLastError();
}

extern "C" int mt_cls_head_bwd_blocks(int B) { return (B + 31) / 32; }

extern "C" int mt_cls_head_bwd_launch(const float* x, const float* gamma, const float* beta, const float* w,
                                      const float* stats, const float* dpred, float* dx, float* part, int B, int d,
                                      hipStream_t stream) {
  if (B <= 0) return 0;
  if (d < 1 || d > 64 * CH_MAXV) return -2;
  hipLaunchKernelGGL(mt_cls_head_bwd_kernel, dim3((unsigned)mt_cls_head_bwd_blocks(B)), dim3(TK_THREADS), 0, stream, x, gamma,
                     beta, w, stats, dpred, dx, part, B, d, 32);
  return (int)hipGetLastError();
}
