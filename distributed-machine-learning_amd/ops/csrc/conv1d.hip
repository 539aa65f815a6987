// Depthwise 1-D convolution over time on channel-last activations [B, S, d] f32 (gfx950), odd kernel
// sizes up to 7, zero padding k/2 -- the temporal half of a depthwise-separable feed-forward block.
//
//   y[b, s, c] = bias[c] + sum_t w[c, t] * x[b, s + t - k/2, c]
//
// PyTorch's Conv1d wants [B, d, S]; the encoder keeps [B, S, d], so the stock path transposes twice per
// call.  Here the kernel works on the channel-last layout directly: a thread owns four consecutive
// channels of one token (16-byte loads/stores, coalesced over channels) and walks the k taps; the
// neighbouring tokens it reads are the lines its neighbours in the grid just touched, so the taps hit L2/L1.
//
// Backward: dx is the same stencil with the taps mirrored; dw/db are reduced per workgroup over a chunk of
// rows into partials[chunk][k + 1][d], which the caller sums (ordered, no atomics).
#include "mt_common.h"

#define CV_MAXK 7
#define CV_ROWS 64      // rows per workgroup in the weight-gradient pass

template <bool MIRROR>
__global__ __launch_bounds__(256) void mt_dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        long rows, int S, int d, int k) {
  const int d4 = d >> 2;
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
  if (i4 >= rows * d4) return;
  const long row = i4 / d4;
  const int c = (int)(i4 % d4) * 4;
  const int s = (int)(row % S);
  const int half = k >> 1;
  f32x4 acc = (bias != nullptr) ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < k; ++t) {
    const int ss = s + t - half;
    if (ss < 0 || ss >= S) continue;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (row + (t - half)) * d + c);
    const int tw = MIRROR ? (k - 1 - t) : t;
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += w[(c + e) * k + tw] * xv[e];
  }
  *reinterpret_cast<f32x4*>(y + row * d + c) = acc;
}

// partials[chunk][k + 1][d]: taps 0..k-1 then the bias gradient.  Block = 64 channels x 4 row lanes.
__global__ __launch_bounds__(256) void mt_dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ partials, long rows, int S, int d,
                                                              int k) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int lane_r = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.y * CV_ROWS;
  const int half = k >> 1;
  float acc[CV_MAXK + 1];
#pragma unroll
  for (int t = 0; t <= CV_MAXK; ++t) acc[t] = 0.f;
  if (c < d) {
    for (long row = r0 + lane_r; row < r0 + CV_ROWS && row < rows; row += 4) {
      const int s = (int)(row % S);
      const float g = dy[row * d + c];
      acc[CV_MAXK] += g;
#pragma unroll
      for (int t = 0; t < CV_MAXK; ++t) {
        const int ss = s + t - half;
        if (t < k && ss >= 0 && ss < S) acc[t] += g * x[(row + (t - half)) * d + c];
      }
    }
  }
  __shared__ float red[4][CV_MAXK + 1][64];
#pragma unroll
  for (int t = 0; t <= CV_MAXK; ++t) red[lane_r][t][threadIdx.x & 63] = acc[t];
  __syncthreads();
  if (lane_r == 0 && c < d) {
    float* out = partials + (size_t)blockIdx.y * (k + 1) * d;
    for (int t = 0; t < k; ++t)
      out[(size_t)t * d + c] = (red[0][t][threadIdx.x] + red[1][t][threadIdx.x]) + (red[2][t][threadIdx.x] + red[3][t][threadIdx.x]);
    out[(size_t)k * d + c] = (red[0][CV_MAXK][threadIdx.x] + red[1][CV_MAXK][threadIdx.x]) +
                             (red[2][CV_MAXK][threadIdx.x] + red[3][CV_MAXK][threadIdx.x]);
  }
}

extern "C" int mt_dwconv_chunks(long rows) { return (int)((rows + CV_ROWS - 1) / CV_ROWS); }

// x, y: [rows = B*S][d]; w: [d][k]; bias: [d] or null.  mirror = 1 computes the input gradient from dy.
extern "C" int mt_dwconv_launch(const float* x, const float* w, const float* bias, float* y, long rows, int S, int d, int k,
                                int mirror, hipStream_t stream) {
  if (d % 4 != 0 || k < 1 || k > CV_MAXK || (k & 1) == 0 || S <= 0 || rows % S != 0) return -2;
  const long total = rows * (d / 4);
  if (total <= 0) return 0;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (mirror) hipLaunchKernelGGL(mt_dwconv_kernel<true>, grid, block, 0, stream, x, w, bias, y, rows, S, d, k);
  else hipLaunchKernelGGL(mt_dwconv_kernel<false>, grid, block, 0, stream, x, w, bias, y, rows, S, d, k);
  return (int)hipGetLastError();
}

extern "C" int mt_dwconv_wgrad_launch(const float* x, const float* dy, float* partials, long rows, int S, int d, int k,
                                      hipStream_t stream) {
  if (k < 1 || k > CV_MAXK || (k & 1) == 0 || S <= 0 || rows % S != 0) return -2;
  if (rows <= 0) return 0;
  dim3 grid((d + 63) / 64, (unsigned)((rows + CV_ROWS - 1) / CV_ROWS)), block(256);
  hipLaunchKernelGGL(mt_dwconv_wgrad_kernel, grid, block, 0, stream, x, dy, partials, rows, S, d, k);
  return (int)hipGetLastError();
}
