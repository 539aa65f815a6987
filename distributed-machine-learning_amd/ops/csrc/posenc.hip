// Positional-encoding add + dropout, forward and backward, for [B, S, d] f32 activations (gfx950).
//
//   y[b, s, :] = keep(b, s, :) * (x[b, s, :] + pe[s, :]) / (1 - p)
//
// The keep mask is never materialised: element i is kept iff hash(seed, i) >= p * 2^32, evaluated again in
// the backward pass.  The seed is read from DEVICE memory so that a step captured in a hipGraph draws a
// fresh mask on every replay (the caller produces the seed with a graph-safe random op).
#include "mt_common.h"

__device__ __forceinline__ uint32_t pe_hash(uint32_t seed, uint32_t i) {
  return mt_mix32(i * 0x9E3779B9u + seed) ^ mt_mix32(seed ^ (i >> 3));
}

__global__ __launch_bounds__(256) void mt_posenc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                            float* __restrict__ y, long total4, int S, int d,
                                                            const int64_t* __restrict__ seed_ptr, uint32_t thresh,
                                                            float scale) {
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
  if (i4 >= total4) return;
  const long i = i4 * 4;
  const int c = (int)(i % d);
  const int s = (int)((i / d) % S);
  f32x4 v = *reinterpret_cast<const f32x4*>(x + i) + *reinterpret_cast<const f32x4*>(pe + (long)s * d + c);
  if (thresh != 0u) {
    const uint32_t seed = (uint32_t)seed_ptr[0];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = pe_hash(seed, (uint32_t)(i + e)) >= thresh ? v[e] * scale : 0.f;
  }
  *reinterpret_cast<f32x4*>(y + i) = v;
}

__global__ __launch_bounds__(256) void mt_posenc_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx,
                                                            long total4, const int64_t* __restrict__ seed_ptr,
                                                            uint32_t thresh, float scale) {
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;
  if (i4 >= total4) return;
  const long i = i4 * 4;
  f32x4 v = *reinterpret_cast<const f32x4*>(dy + i);
  if (thresh != 0u) {
    const uint32_t seed = (uint32_t)seed_ptr[0];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = pe_hash(seed, (uint32_t)(i + e)) >= thresh ? v[e] * scale : 0.f;
  }
  *reinterpret_cast<f32x4*>(dx + i) = v;
}

// x, y: [rows = B*S][d] f32 contiguous, d % 4 == 0; pe: [>= S][d]; p in [0, 1).
extern "C" int mt_posenc_fwd_launch(const float* x, const float* pe, float* y, long rows, int S, int d,
                                    const int64_t* seed, float p, hipStream_t stream) {
  if (d % 4 != 0 || S <= 0) return -2;
  if (p > 0.f && seed == nullptr) return -3;
  const long total4 = rows * d / 4;
  if (total4 <= 0) return 0;
  const uint32_t thresh = p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u;
  hipLaunchKernelGGL(mt_posenc_fwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, x, pe, y, total4,
                     S, d, seed, thresh, p > 0.f ? 1.f / (1.f - p) : 1.f);
  return (int)hipGetLastError();
}

extern "C" int mt_posenc_bwd_launch(const float* dy, float* dx, long rows, int d, const int64_t* seed, float p,
                                    hipStream_t stream) {
  if (d % 4 != 0) return -2;
  if (p > 0.f && seed == nullptr) return -3;
  const long total4 = rows * d / 4;
  if (total4 <= 0) return 0;
  const uint32_t thresh = p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u;
  hipLaunchKernelGGL(mt_posenc_bwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, stream, dy, dx, total4,
                     seed, thresh, p > 0.f ? 1.f / (1.f - p) : 1.f);
  return (int)hipGetLastError();
}
